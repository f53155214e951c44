/*
 * rb_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * Plain-C CPU restatement of the alshai/rowbowt rb_align hot path.  See rb_oracle.h for
 * the parity status (PINNED against the reference's own fixtures + golden values).
 *
 * Shape: the reference's call chain is kept (RowBowt::LF -> rle_string::rank -> runs /
 * runs_per_letter / run_heads), with each sdsl primitive restated over a plain decoded array:
 *   sdsl::sd_vector<> + rank_1/select_1   -> sorted array of one-positions (sdv_t)
 *   sdsl::wt_huff<>                        -> heads[R] + per-symbol sorted run-index lists (hs_t)
 *   sdsl::int_vector<>                     -> uint64_t[]
 * (orc_set_reference_shaped switches the first two to Elias-Fano vectors and a Huffman-shaped wavelet tree: the structures
 * sdsl holds, restated from their published layouts -- same answers, the reference's memory behaviour)
 * The sdsl on-disk layout read here is the one the reference's shipped fixtures use
 * (SURVEY.md section 8b-format); the reader consumes each fixture to exact EOF or fails.
 */
#define _POSIX_C_SOURCE 200809L
#include "rb_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* byte reader                                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t *buf;
    size_t len, pos;
    int err;
} rd_t;

static int rd_open(rd_t *r, const char *fname) {
    memset(r, 0, sizeof(*r));
    FILE *f = fopen(fname, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    r->buf = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
    r->len = (size_t)sz;
    if (fread(r->buf, 1, r->len, f) != r->len) { fclose(f); free(r->buf); return -1; }
    fclose(f);
    return 0;
}
static void rd_close(rd_t *r) { free(r->buf); r->buf = NULL; }
static void rd_bytes(rd_t *r, void *dst, size_t n) {
    if (r->err || r->pos + n > r->len) { r->err = 1; memset(dst, 0, n); return; }
    memcpy(dst, r->buf + r->pos, n);
    r->pos += n;
}
static void rd_skip(rd_t *r, size_t n) {
    if (r->err || r->pos + n > r->len) { r->err = 1; return; }
    r->pos += n;
}
static uint64_t rd_u64(rd_t *r) { uint64_t v; rd_bytes(r, &v, 8); return v; }
static uint8_t rd_u8(rd_t *r) { uint8_t v; rd_bytes(r, &v, 1); return v; }

/* sdsl int_vector<w> as serialised in the fixtures: one u64 header, low 56 bits = length in
 * bits, top byte = element width; then ceil(bits/64) little-endian words, LSB-first packing. */
typedef struct {
    uint64_t bits, len;
    unsigned width;
    uint64_t *w; /* owned */
    uint64_t nwords;
} iv_t;

static void iv_read(rd_t *r, iv_t *v, int keep) {
    uint64_t h = rd_u64(r);
    v->bits = h & ((1ULL << 56) - 1);
    v->width = (unsigned)(h >> 56);
    if (v->width == 0 || v->width > 64) { r->err = 1; v->width = 1; }
    v->len = v->bits / v->width;
    v->nwords = (v->bits + 63) / 64;
    v->w = NULL;
    if (keep) {
        if (r->err || r->pos + v->nwords * 8 > r->len) { r->err = 1; v->nwords = 0; v->len = 0; return; }
        v->w = (uint64_t *)malloc((v->nwords + 1) * 8);
        rd_bytes(r, v->w, v->nwords * 8);
        v->w[v->nwords] = 0;
    } else {
        rd_skip(r, v->nwords * 8);
    }
}
static inline uint64_t iv_get(const iv_t *v, uint64_t i) {
    uint64_t b = i * v->width, wi = b >> 6;
    unsigned sh = (unsigned)(b & 63);
    uint64_t x = v->w[wi] >> sh;
    if (sh + v->width > 64) x |= v->w[wi + 1] << (64 - sh);
    return v->width == 64 ? x : (x & ((1ULL << v->width) - 1));
}
static void iv_free(iv_t *v) { free(v->w); v->w = NULL; }

/* sdsl::select_support_mcl<>: skipped, never used (positions are decoded outright). */
static void skip_select_mcl(rd_t *r) {
    uint64_t arg_cnt = rd_u64(r);
    if (arg_cnt == 0) return;
    iv_t t;
    iv_read(r, &t, 0);            /* superblock */
    iv_read(r, &t, 0);            /* mini_or_long */
    uint64_t sb = (arg_cnt + 4095) >> 12;
    for (uint64_t i = 0; i < sb && !r->err; ++i) iv_read(r, &t, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* sparse_sd_vector restated: sorted one-positions                                            */
/* ------------------------------------------------------------------------------------------ */
/* REFERENCE-SHAPED MODE (orc_set_reference_shaped; SURVEY 8d's optional CPU mode).  The plain arrays above answer rank by
 * binary search -- not what sdsl does.  With `ef` set an sdv_t answers from the structure sdsl::sd_vector<> holds (Elias-Fano:
 * the low wl bits of every position packed in `low`, the high parts in unary in the bit-vector `high`, one sampled position per
 * 64 ones / 64 zeros standing in for select_support_mcl), rank as rank_support_sd does it (select_0 of the high part, then a
 * scan of the bucket) and select as select_support_sd does (select_1 on `high`, low bits appended): the memory behaviour of
 * the reference's rle_string::rank, restated from the published layout, not from sdsl's sources (absent here). */
typedef struct {
    unsigned wl;
    uint64_t *low;        /* m entries of wl bits, LSB-first */
    uint64_t *high;       /* m ones, (u >> wl) + 1 zeros: bucket b = its ones, then a zero */
    uint64_t high_bits;
    uint64_t *sel1, *sel0; /* position in `high` of one / zero number 64 j */
} ef_t;
typedef struct {
    uint64_t u;    /* universe (size()) */
    uint64_t m;    /* number_of_1() */
    uint64_t *ones;
    ef_t *ef;      /* reference-shaped mode, else NULL */
} sdv_t;

static inline uint64_t ef_low(const ef_t *e, uint64_t i) {
    if (!e->wl) return 0;
    const uint64_t bit = i * e->wl, w = bit >> 6, o = bit & 63;
    uint64_t v = e->low[w] >> o;
    if (o + e->wl > 64) v |= e->low[w + 1] << (64 - o);
    return v & ((UINT64_C(1) << e->wl) - 1);
}
/* position of the r-th (0-based) set bit of w (r < popcount(w)) */
static inline unsigned word_select(uint64_t w, unsigned r) {
    unsigned base = 0;
    for (;;) {   /* a byte at a time, then bit by bit */
        const unsigned c = (unsigned)__builtin_popcount((unsigned)(w & 0xFF));
        if (r < c) break;
        r -= c; w >>= 8; base += 8;
    }
    while (r--) w &= w - 1;
    return base + (unsigned)__builtin_ctzll(w);
}
/* position in `high` of one number i (ones != 0) or zero number i (ones == 0) */
static inline uint64_t ef_select_high(const ef_t *e, uint64_t i, int ones) {
    uint64_t p = (ones ? e->sel1 : e->sel0)[i >> 6];
    unsigned r = (unsigned)(i & 63);
    uint64_t wi = p >> 6;
    uint64_t w = (ones ? e->high[wi] : ~e->high[wi]) & (~UINT64_C(0) << (p & 63));
    for (;;) {
        const unsigned c = (unsigned)__builtin_popcountll(w);
        if (r < c) return (wi << 6) + word_select(w, r);
        r -= c;
        ++wi;
        w = ones ? e->high[wi] : ~e->high[wi];
    }
}
static void ef_build(sdv_t *s) {
    if (s->ef || s->m == 0) return;
    ef_t *e = (ef_t *)calloc(1, sizeof(ef_t));
    uint64_t q = s->u / s->m;
    e->wl = 0;
    while ((q >> 1) >= 1) { q >>= 1; e->wl++; }   /* floor(log2(u / m)): sd_vector's split between low and high bits */
    e->low = (uint64_t *)calloc((s->m * e->wl + 63) / 64 + 2, 8);
    e->high_bits = s->m + (s->u >> e->wl) + 2;
    e->high = (uint64_t *)calloc(e->high_bits / 64 + 2, 8);
    for (uint64_t i = 0; i < s->m; ++i) {
        const uint64_t v = s->ones[i], hp = (v >> e->wl) + i;
        e->high[hp >> 6] |= UINT64_C(1) << (hp & 63);
        if (e->wl) {
            const uint64_t lo = v & ((UINT64_C(1) << e->wl) - 1), bit = i * e->wl, w = bit >> 6, o = bit & 63;
            e->low[w] |= lo << o;
            if (o + e->wl > 64) e->low[w + 1] |= lo >> (64 - o);
        }
    }
    const uint64_t nz = e->high_bits - s->m;
    e->sel1 = (uint64_t *)malloc((s->m / 64 + 2) * 8);
    e->sel0 = (uint64_t *)malloc((nz / 64 + 2) * 8);
    uint64_t k1 = 0, k0 = 0;
    for (uint64_t p = 0; p < e->high_bits; ++p) {
        if ((e->high[p >> 6] >> (p & 63)) & 1) { if ((k1 & 63) == 0) e->sel1[k1 >> 6] = p; ++k1; }
        else { if ((k0 & 63) == 0) e->sel0[k0 >> 6] = p; ++k0; }
    }
    s->ef = e;
}
static void ef_free(sdv_t *s) {
    if (!s->ef) return;
    free(s->ef->low); free(s->ef->high); free(s->ef->sel1); free(s->ef->sel0); free(s->ef);
    s->ef = NULL;
}

/* sdsl::sd_vector<>::load: size, wl, low, high, select_1 support, select_0 support */
static void sdv_read_raw(rd_t *r, sdv_t *s) {
    s->ef = NULL;
    uint64_t size = rd_u64(r);
    unsigned wl = rd_u8(r);
    iv_t low, high;
    iv_read(r, &low, 1);
    iv_read(r, &high, 1);
    skip_select_mcl(r);
    skip_select_mcl(r);
    s->u = size;
    s->m = low.len;
    s->ones = (uint64_t *)malloc((s->m + 1) * 8);
    uint64_t k = 0;
    for (uint64_t p = 0; p < high.bits && k < s->m && !r->err; ++p) {
        if ((high.w[p >> 6] >> (p & 63)) & 1) {
            s->ones[k] = ((p - k) << wl) | (wl ? iv_get(&low, k) : 0);
            ++k;
        }
    }
    if (k != s->m) r->err = 1;
    iv_free(&low);
    iv_free(&high);
}
/* sparse_sd_vector.hpp:194-200 load */
static void sdv_read(rd_t *r, sdv_t *s) {
    s->u = rd_u64(r);
    s->m = 0;
    s->ones = NULL;
    s->ef = NULL;
    if (s->u == 0) return;
    sdv_read_raw(r, s);
}
/* sparse_sd_vector.hpp:110-113: ones in [0,i) */
static inline uint64_t sdv_rank(const sdv_t *s, uint64_t i) {
    if (s->ef) {   /* rank_support_sd: the bucket of i's high part starts behind zero number hi - 1; scan its ones */
        const ef_t *e = s->ef;
        if (i >= s->u) return s->m;
        const uint64_t hb = i >> e->wl, lb = e->wl ? (i & ((UINT64_C(1) << e->wl) - 1)) : 0;
        uint64_t p = hb ? ef_select_high(e, hb - 1, 0) + 1 : 0;   /* first bit of bucket hb */
        uint64_t idx = p - hb;                                      /* ones before it */
        while (((e->high[p >> 6] >> (p & 63)) & 1) && ef_low(e, idx) < lb) { ++p; ++idx; }
        return idx;
    }
    uint64_t lo = 0, hi = s->m;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (s->ones[mid] < i) lo = mid + 1; else hi = mid;
    }
    return lo;
}
/* sparse_sd_vector.hpp:160-163: 0-based select */
static inline uint64_t sdv_select(const sdv_t *s, uint64_t i) {
    if (s->ef) return ((ef_select_high(s->ef, i, 1) - i) << s->ef->wl) | ef_low(s->ef, i);   /* select_support_sd */
    return s->ones[i];
}
/* sparse_sd_vector.hpp:150-154 */
static inline uint64_t sdv_gapAt(const sdv_t *s, uint64_t i) {
    if (i == 0) return sdv_select(s, 0) + 1;
    return sdv_select(s, i) - sdv_select(s, i - 1);
}
/* sparse_sd_vector.hpp:141-143 */
static inline uint64_t sdv_pred_rank_circular(const sdv_t *s, uint64_t i) {
    uint64_t rk = sdv_rank(s, i);
    return rk == 0 ? s->m - 1 : rk - 1;
}
static void sdv_free(sdv_t *s) { ef_free(s); free(s->ones); s->ones = NULL; }

/* ------------------------------------------------------------------------------------------ */
/* huff_string restated: heads[] + per-symbol sorted lists of run indices                     */
/* ------------------------------------------------------------------------------------------ */
/* Reference-shaped mode: a Huffman-shaped wavelet tree over the run heads (sdsl::wt_huff<> behind huff_string): one bit per
 * head and tree level in a single bit-vector, rank by 512-bit blocks (a cumulative count per block + popcounts inside it, as
 * rank_support_v), access / rank top-down along the symbol's code, select bottom-up by binary search over ranks. */
typedef struct {
    uint64_t off, len, ones_before;   /* this node's stretch of `bv`; ones of `bv` before it */
    int child[2];                     /* node index, or -1 - symbol for a leaf */
    int parent, parent_bit;
} wtn_t;
typedef struct {
    uint64_t *bv, *blk;   /* blk[j] = ones in bv[0, 512 j) */
    uint64_t bits;
    wtn_t *node;
    int nnodes, root;     /* root < 0: a single symbol (-1 - symbol) */
    int leaf_parent[256], leaf_bit[256];
    uint32_t code[256];   /* bits from the root, first step in bit 0 */
    uint8_t code_len[256];
} wt_t;
typedef struct {
    uint64_t size;
    uint8_t *heads;
    uint64_t cnt[256];
    uint64_t *pos[256];
    wt_t *wt;             /* reference-shaped mode, else NULL */
} hs_t;

static inline uint64_t wt_rank1(const wt_t *t, uint64_t p) {   /* ones in bv[0, p) */
    uint64_t r = t->blk[p >> 9];
    for (uint64_t w = (p >> 9) << 3; w < (p >> 6); ++w) r += (uint64_t)__builtin_popcountll(t->bv[w]);
    if (p & 63) r += (uint64_t)__builtin_popcountll(t->bv[p >> 6] & ((UINT64_C(1) << (p & 63)) - 1));
    return r;
}
static void wt_build(hs_t *h) {
    if (h->wt || h->size == 0) return;
    wt_t *t = (wt_t *)calloc(1, sizeof(wt_t));
    /* Huffman tree over the symbol counts (ties by symbol value: any prefix code gives the same answers) */
    int alive[512], na = 0;
    uint64_t wgt[512];
    t->node = (wtn_t *)calloc(256, sizeof(wtn_t));
    for (int c = 0; c < 256; ++c)
        if (h->cnt[c]) { alive[na] = -1 - c; wgt[na] = h->cnt[c]; ++na; }
    for (int c = 0; c < 256; ++c) { t->leaf_parent[c] = -1; t->leaf_bit[c] = 0; }
    if (na == 1) { t->root = alive[0]; h->wt = t; return; }
    while (na > 1) {
        int a = 0, b = 1;
        if (wgt[b] < wgt[a]) { int x = a; a = b; b = x; }
        for (int j = 2; j < na; ++j) {
            if (wgt[j] < wgt[a]) { b = a; a = j; }
            else if (wgt[j] < wgt[b]) b = j;
        }
        const int id = t->nnodes++;
        t->node[id].child[0] = alive[a];
        t->node[id].child[1] = alive[b];
        t->node[id].parent = -1;
        for (int k = 0; k < 2; ++k) {
            const int ch = t->node[id].child[k];
            if (ch >= 0) { t->node[ch].parent = id; t->node[ch].parent_bit = k; }
            else { t->leaf_parent[-1 - ch] = id; t->leaf_bit[-1 - ch] = k; }
        }
        const uint64_t w = wgt[a] + wgt[b];
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        alive[lo] = id; wgt[lo] = w;
        alive[hi] = alive[na - 1]; wgt[hi] = wgt[na - 1];
        --na;
    }
    t->root = alive[0];
    /* codes */
    for (int c = 0; c < 256; ++c) {
        if (!h->cnt[c]) continue;
        uint32_t code = 0; int len = 0;   /* walked leaf -> root: every step pushes the earlier ones up, so bit 0 ends as the root's */
        int nd = t->leaf_parent[c], bit = t->leaf_bit[c];
        while (nd >= 0) { code = (code << 1) | (uint32_t)bit; ++len; bit = t->node[nd].parent_bit; nd = t->node[nd].parent; }
        t->code[c] = code;
        t->code_len[c] = (uint8_t)len;
    }
    /* node lengths: heads that pass through the node */
    for (int id = 0; id < t->nnodes; ++id) t->node[id].len = 0;
    for (int c = 0; c < 256; ++c) {
        if (!h->cnt[c]) continue;
        int nd = t->root;
        for (int j = 0; j < t->code_len[c]; ++j) { t->node[nd].len += h->cnt[c]; const int ch = t->node[nd].child[(t->code[c] >> j) & 1]; if (ch < 0) break; nd = ch; }
    }
    uint64_t off = 0;
    for (int id = 0; id < t->nnodes; ++id) { t->node[id].off = off; off += t->node[id].len; }
    t->bits = off;
    t->bv = (uint64_t *)calloc(off / 64 + 2, 8);
    uint64_t *fill = (uint64_t *)calloc((size_t)t->nnodes + 1, 8);
    for (uint64_t i = 0; i < h->size; ++i) {
        const uint8_t c = h->heads[i];
        int nd = t->root;
        for (int j = 0; j < t->code_len[c]; ++j) {
            const int b = (int)((t->code[c] >> j) & 1);
            const uint64_t p = t->node[nd].off + fill[nd]++;
            if (b) t->bv[p >> 6] |= UINT64_C(1) << (p & 63);
            const int ch = t->node[nd].child[b];
            if (ch < 0) break;
            nd = ch;
        }
    }
    free(fill);
    t->blk = (uint64_t *)malloc((off / 512 + 2) * 8);
    uint64_t acc = 0;
    for (uint64_t w = 0; w <= off / 64 + 1; ++w) {
        if ((w & 7) == 0) t->blk[w >> 3] = acc;
        acc += (uint64_t)__builtin_popcountll(t->bv[w]);
    }
    for (int id = 0; id < t->nnodes; ++id) t->node[id].ones_before = wt_rank1(t, t->node[id].off);
    h->wt = t;
}
static void wt_free(hs_t *h) {
    if (!h->wt) return;
    free(h->wt->bv); free(h->wt->blk); free(h->wt->node); free(h->wt);
    h->wt = NULL;
}

static void hs_index(hs_t *h) {
    memset(h->cnt, 0, sizeof(h->cnt));
    for (uint64_t i = 0; i < h->size; ++i) h->cnt[h->heads[i]]++;
    for (int c = 0; c < 256; ++c) h->pos[c] = h->cnt[c] ? (uint64_t *)malloc(h->cnt[c] * 8) : NULL;
    uint64_t fill[256];
    memset(fill, 0, sizeof(fill));
    for (uint64_t i = 0; i < h->size; ++i) { uint8_t c = h->heads[i]; h->pos[c][fill[c]++] = i; }
}
/* huff_string.hpp:30-33 */
static inline uint8_t hs_at(const hs_t *h, uint64_t i) {
    if (h->wt) {   /* wt_huff::operator[]: down the tree, the position mapped by rank at every level */
        const wt_t *t = h->wt;
        int nd = t->root;
        while (nd >= 0) {
            const wtn_t *q = &t->node[nd];
            const uint64_t p = q->off + i, r1 = wt_rank1(t, p) - q->ones_before;
            const int b = (int)((t->bv[p >> 6] >> (p & 63)) & 1);
            i = b ? r1 : i - r1;
            nd = q->child[b];
        }
        return (uint8_t)(-1 - nd);
    }
    return h->heads[i];
}
/* huff_string.hpp:39-42: number of c in heads[0,i) */
static inline uint64_t hs_rank(const hs_t *h, uint64_t i, uint8_t c) {
    if (h->wt) {   /* wt_huff::rank: along c's code */
        const wt_t *t = h->wt;
        if (!h->cnt[c]) return 0;
        int nd = t->root;
        for (int j = 0; nd >= 0; ++j) {
            const wtn_t *q = &t->node[nd];
            const uint64_t r1 = wt_rank1(t, q->off + i) - q->ones_before;
            const int b = (int)((t->code[c] >> j) & 1);
            i = b ? r1 : i - r1;
            nd = q->child[b];
        }
        return i;
    }
    const uint64_t *p = h->pos[c];
    uint64_t lo = 0, hi = h->cnt[c];
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (p[mid] < i) lo = mid + 1; else hi = mid;
    }
    return lo;
}
/* huff_string.hpp:47-49: 0-based select */
static inline uint64_t hs_select(const hs_t *h, uint64_t i, uint8_t c) {
    if (h->wt) {   /* wt_huff::select: from c's leaf up, at every node the position of the (i + 1)-th bit of the branch taken */
        const wt_t *t = h->wt;
        int nd = t->leaf_parent[c], b = t->leaf_bit[c];
        while (nd >= 0) {
            const wtn_t *q = &t->node[nd];
            uint64_t lo = 0, hi = q->len;   /* smallest x with (# b-bits in [0, x]) == i + 1 */
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                const uint64_t r1 = wt_rank1(t, q->off + mid + 1) - q->ones_before;
                const uint64_t cb = b ? r1 : mid + 1 - r1;
                if (cb < i + 1) lo = mid + 1; else hi = mid;
            }
            i = lo;
            b = q->parent_bit;
            nd = q->parent;
        }
        return i;
    }
    return h->pos[c][i];
}
static void hs_free(hs_t *h) {
    wt_free(h);
    free(h->heads);
    for (int c = 0; c < 256; ++c) free(h->pos[c]);
}

/* sdsl::wt_huff<>::load, decoded outright into heads[] (huff_string.hpp:61-63).
 * Layout: size, sigma, bv, rank_support_v (int_vector<64>), select_1, select_0, n_nodes,
 * n_nodes x 22-byte {u64 bv_pos, u64 bv_pos_rank, u16 parent, u16 child0, u16 child1},
 * u16 c_to_leaf[256], u64 path[256]. */
typedef struct { uint64_t bv_pos, bv_pos_rank; uint16_t parent, child[2]; } wtnode_t;

static void wt_expand(const iv_t *bv, const wtnode_t *nodes, const int *leaf_sym, uint64_t n_nodes,
                      uint16_t v, uint64_t *idx, uint64_t cnt, uint64_t *scratch, uint8_t *out, int *err) {
    if (*err || cnt == 0) return;
    if (v >= n_nodes) { *err = 1; return; }
    if (leaf_sym[v] >= 0) {
        for (uint64_t i = 0; i < cnt; ++i) out[idx[i]] = (uint8_t)leaf_sym[v];
        return;
    }
    uint64_t base = nodes[v].bv_pos;
    if (base + cnt > bv->bits) { *err = 1; return; }
    uint64_t n0 = 0, n1 = 0;
    for (uint64_t i = 0; i < cnt; ++i) {
        uint64_t p = base + i;
        if ((bv->w[p >> 6] >> (p & 63)) & 1) scratch[n1++] = idx[i]; else idx[n0++] = idx[i];
    }
    memcpy(idx + n0, scratch, n1 * 8);
    wt_expand(bv, nodes, leaf_sym, n_nodes, nodes[v].child[0], idx, n0, scratch, out, err);
    wt_expand(bv, nodes, leaf_sym, n_nodes, nodes[v].child[1], idx + n0, n1, scratch, out, err);
}

static void hs_read(rd_t *r, hs_t *h) {
    memset(h, 0, sizeof(*h));
    uint64_t size = rd_u64(r);
    uint64_t sigma = rd_u64(r);
    (void)sigma;
    iv_t bv, t;
    iv_read(r, &bv, 1);
    iv_read(r, &t, 0); /* rank_support_v */
    skip_select_mcl(r);
    skip_select_mcl(r);
    uint64_t n_nodes = rd_u64(r);
    if (n_nodes > 65535) { r->err = 1; n_nodes = 0; }
    wtnode_t *nodes = (wtnode_t *)calloc(n_nodes + 1, sizeof(wtnode_t));
    for (uint64_t i = 0; i < n_nodes; ++i) {
        uint8_t raw[22];
        rd_bytes(r, raw, 22);
        memcpy(&nodes[i].bv_pos, raw, 8);
        memcpy(&nodes[i].bv_pos_rank, raw + 8, 8);
        memcpy(&nodes[i].parent, raw + 16, 2);
        memcpy(&nodes[i].child[0], raw + 18, 2);
        memcpy(&nodes[i].child[1], raw + 20, 2);
    }
    uint16_t c_to_leaf[256];
    rd_bytes(r, c_to_leaf, sizeof(c_to_leaf));
    rd_skip(r, 256 * 8); /* path[256] */
    int *leaf_sym = (int *)malloc((n_nodes + 1) * sizeof(int));
    for (uint64_t i = 0; i <= n_nodes; ++i) leaf_sym[i] = -1;
    for (int c = 0; c < 256; ++c)
        if (c_to_leaf[c] != 0xFFFF && c_to_leaf[c] < n_nodes) leaf_sym[c_to_leaf[c]] = c;
    h->size = size;
    h->heads = (uint8_t *)malloc(size ? size : 1);
    if (!r->err && size) {
        uint64_t *idx = (uint64_t *)malloc(size * 8), *scratch = (uint64_t *)malloc(size * 8);
        for (uint64_t i = 0; i < size; ++i) idx[i] = i;
        int err = 0;
        wt_expand(&bv, nodes, leaf_sym, n_nodes, 0, idx, size, scratch, h->heads, &err);
        if (err) r->err = 1;
        free(idx);
        free(scratch);
    }
    free(leaf_sym);
    free(nodes);
    iv_free(&bv);
    if (!r->err) hs_index(h);
}

/* ------------------------------------------------------------------------------------------ */
/* the index                                                                                  */
/* ------------------------------------------------------------------------------------------ */
struct orc_index {
    /* ri::rle_string members, rle_string.hpp:381-392 */
    uint64_t B, n, R;
    sdv_t runs;
    sdv_t rpl[256]; /* runs_per_letter */
    hs_t run_heads;
    /* RowBowt::f_, rowbowt.hpp:789 (one extra slot so f[c+1] is defined for c==255) */
    uint64_t f[257];
    /* ToeholdSA members, toehold_sa.hpp:157-161 */
    int has_tsa;
    uint64_t tsa_r, tsa_n;
    sdv_t pred;
    uint64_t *samples_last;
    uint64_t *pred_to_run;
    /* MarkerArray (pfbwt-f, un-vendored; layout per SURVEY 8b-format) */
    int has_ma;
    uint64_t ma_nruns, ma_nvals;
    uint64_t *ma_start, *ma_end, *ma_off, *ma_vals;
    int32_t ma_wsize;
    /* DocList, doclist.hpp:81-82 */
    int has_dl;
    uint64_t ndocs;
    char **doc_names;
    uint64_t *doc_starts;
    uint64_t *doc_sorted;
};

/* rowbowt.hpp:770-778 */
static void build_f(orc_index *x) {
    memset(x->f, 0, sizeof(x->f));
    uint64_t p = 0;
    for (int i = 0; i < 256; ++i) {
        p += x->rpl[i].u; /* rank(size(), i) == runs_per_letter[i].size(), rle_string.hpp:135 */
        x->f[i + 1] = p;
    }
}

/* rle_string.hpp:238-242 */
static inline uint64_t run_at(const orc_index *x, uint64_t i) {
    uint8_t c = hs_at(&x->run_heads, i);
    return sdv_gapAt(&x->rpl[c], hs_rank(&x->run_heads, i, c));
}

/* rle_string.hpp:131-161 */
uint64_t orc_rank(const orc_index *x, uint64_t i, uint8_t c) {
    if (x->rpl[c].u == 0) return 0;
    if (i == x->n) return x->rpl[c].u;
    uint64_t last_block = sdv_rank(&x->runs, i);
    uint64_t current_run = last_block * x->B;
    uint64_t pos = 0;
    if (last_block > 0) pos = sdv_select(&x->runs, last_block - 1) + 1;
    uint64_t dist = i - pos;
    while (pos < i) {
        pos += run_at(x, current_run);
        current_run++;
        if (pos <= i) dist = i - pos;
    }
    if (pos > i) current_run--;
    uint64_t rk = hs_rank(&x->run_heads, current_run, c);
    uint64_t tail = (hs_at(&x->run_heads, current_run) == c) * dist;
    if (rk == 0) return tail;
    return sdv_select(&x->rpl[c], rk - 1) + 1 + tail;
}

/* rle_string.hpp:107-126 */
uint64_t orc_select(const orc_index *x, uint64_t i, uint8_t c) {
    uint64_t j = sdv_rank(&x->rpl[c], i);
    uint64_t before = (j == 0 ? i : i - (sdv_select(&x->rpl[c], j - 1) + 1));
    uint64_t r = hs_select(&x->run_heads, j, c);
    uint64_t k = (r / x->B == 0 ? 0 : sdv_select(&x->runs, r / x->B - 1) + 1);
    for (uint64_t t = (r / x->B) * x->B; t < r; ++t) k += run_at(x, t);
    return k + before;
}

/* rle_string.hpp:356-377 run_of: <run containing i, last position of that run> */
static void run_of(const orc_index *x, uint64_t i, uint64_t *run, uint64_t *last) {
    uint64_t last_block = sdv_rank(&x->runs, i);
    uint64_t current_run = last_block * x->B;
    uint64_t pos = 0;
    if (last_block > 0) pos = sdv_select(&x->runs, last_block - 1) + 1;
    while (pos < i) {
        pos += run_at(x, current_run);
        current_run++;
    }
    if (pos > i) current_run--;
    else pos += run_at(x, current_run);
    *run = current_run;
    *last = pos - 1;
}

/* rle_string.hpp:99-102 */
uint8_t orc_access(const orc_index *x, uint64_t i) {
    uint64_t run, last;
    run_of(x, i, &run, &last);
    return hs_at(&x->run_heads, run);
}

/* rle_string.hpp:166-186 */
uint64_t orc_run_of_position(const orc_index *x, uint64_t i) {
    uint64_t last_block = sdv_rank(&x->runs, i);
    uint64_t current_run = last_block * x->B;
    uint64_t pos = 0;
    if (last_block > 0) pos = sdv_select(&x->runs, last_block - 1) + 1;
    while (pos < i) {
        pos += run_at(x, current_run);
        current_run++;
    }
    if (pos > i) current_run--;
    return current_run;
}

/* toehold_sa.hpp:56-72 */
uint64_t orc_phi(const orc_index *x, uint64_t i) {
    uint64_t jr = sdv_pred_rank_circular(&x->pred, i);
    uint64_t j = sdv_select(&x->pred, jr);
    uint64_t delta = j < i ? i - j : i + 1;
    /* the reference asserts pred_to_run_[jr] > 0 (:67): phi(SA[0]) is outside its domain.  It is reached
     * only through a toehold that wrapped below zero (a match at text position 0); 0 stands in there so
     * that this checker never reads out of bounds */
    uint64_t run = x->pred_to_run[jr];
    uint64_t prev_sample = run ? x->samples_last[run - 1] : 0;
    return (prev_sample + delta) % x->tsa_n;
}

/* toehold_sa.hpp:97-99 */
uint64_t orc_last_run_sample(const orc_index *x) {
    return (x->samples_last[x->tsa_r - 1] + 1) % x->tsa_n;
}

/* rowbowt.hpp:74-88.  For c==255 the reference reads f_[256] (past the 256-entry vector) when
 * symbol 255 occurs; here f has 257 entries with f[256]==n, which is the evident intent. */
void orc_LF(const orc_index *x, uint64_t lo, uint64_t hi, uint8_t c, uint64_t *lo_out, uint64_t *hi_out) {
    if ((c == 255 && x->f[c] == x->n) || x->f[c] >= x->f[c + 1]) { *lo_out = 1; *hi_out = 0; return; }
    uint64_t c_before = orc_rank(x, lo, c);
    uint64_t c_inside = orc_rank(x, hi + 1, c) - c_before;
    if (c_inside == 0) { *lo_out = 1; *hi_out = 0; return; }
    uint64_t l = x->f[c] + c_before;
    *lo_out = l;
    *hi_out = l + c_inside - 1;
}

/* rowbowt.hpp:121-131 (ft_ never set on the rb_align path) */
void orc_find_range(const orc_index *x, const uint8_t *q, uint64_t m, uint64_t *lo, uint64_t *hi) {
    uint64_t l = 0, h = x->n - 1; /* full_range(), rowbowt.hpp:115-118 */
    for (uint64_t i = 0; i < m && h >= l; ++i) orc_LF(x, l, h, q[m - i - 1], &l, &h);
    *lo = l;
    *hi = h;
}

/* rowbowt.hpp:555-573 */
static void LF_w_loc(const orc_index *x, uint64_t lo, uint64_t hi, uint8_t c, uint64_t k,
                     uint64_t *nlo, uint64_t *nhi, uint64_t *nk) {
    orc_LF(x, lo, hi, c, nlo, nhi);
    if (*nlo <= *nhi) {
        if (orc_access(x, hi) == c) {
            *nk = k - 1;
        } else {
            uint64_t rnk = orc_rank(x, hi, c) - 1;
            uint64_t j = orc_select(x, rnk, c);
            uint64_t run_of_j = orc_run_of_position(x, j);
            *nk = x->samples_last[run_of_j]; /* toehold_sa.hpp:93-95 */
        }
    } else {
        *nlo = 1; *nhi = 0; *nk = 0;
    }
}

/* rowbowt.hpp:169-184 */
void orc_find_range_w_toehold(const orc_index *x, const uint8_t *q, uint64_t m, uint64_t *lo, uint64_t *hi, uint64_t *ssamp) {
    if (!x->has_tsa) { *lo = 1; *hi = 0; *ssamp = 0; return; }
    uint64_t l = 0, h = x->n - 1, k = orc_last_run_sample(x);
    for (uint64_t i = 0; i < m; ++i) {
        LF_w_loc(x, l, h, q[m - i - 1], k, &l, &h, &k);
        if (h < l) { *lo = 1; *hi = 0; *ssamp = 0; return; } /* LFData::clear, rowbowt.hpp:153-159 */
    }
    *lo = l; *hi = h; *ssamp = k;
}

/* toehold_sa.hpp:37-49 */
uint64_t orc_locs_at(const orc_index *x, uint64_t lo, uint64_t hi, uint64_t k, uint64_t max_hits, uint64_t *out) {
    uint64_t n_occ = hi >= lo ? (hi - lo) + 1 : 0;
    if (n_occ > max_hits) n_occ = max_hits;
    uint64_t k1 = k;
    if (n_occ > 0) {
        out[0] = k1;
        for (uint64_t i = 1; i < n_occ; ++i) {
            k1 = orc_phi(x, k1);
            out[i] = k1;
        }
    }
    return n_occ;
}

/* MarkerArray::at_range (pfbwt-f, un-vendored): concatenation, in run order,
 * of the value lists of all runs with start <= hi && end >= lo. */
uint64_t orc_markers_at(const orc_index *x, uint64_t lo, uint64_t hi, uint64_t *out) {
    if (!x->has_ma || hi < lo) return 0;
    uint64_t cnt = 0;
    /* first run with end >= lo */
    uint64_t a = 0, b = x->ma_nruns;
    while (a < b) { uint64_t mid = (a + b) >> 1; if (x->ma_end[mid] < lo) a = mid + 1; else b = mid; }
    for (uint64_t j = a; j < x->ma_nruns && x->ma_start[j] <= hi; ++j) {
        for (uint64_t t = x->ma_off[j]; t < x->ma_off[j + 1]; ++t) {
            if (out) out[cnt] = x->ma_vals[t];
            ++cnt;
        }
    }
    return cnt;
}

/* rowbowt.hpp:292-339.  markers are *prepended* per window (:320,:333). */
uint64_t orc_find_range_w_markers(const orc_index *x, const uint8_t *q, uint64_t m, uint64_t wsize, uint64_t max_range,
                                  uint64_t *lo, uint64_t *hi, uint64_t *out, uint64_t cap) {
    *lo = 1; *hi = 0;
    if (!x->has_ma) return 0;
    if (m < wsize) return 0;
    uint64_t l = 0, h = x->n - 1;
    uint64_t window_ei = m;
    uint64_t total = 0;
    uint64_t *acc = NULL, acc_cap = 0; /* lf.markers, kept in final order */
#define PREPEND_QUERY()                                                                       \
    do {                                                                                      \
        if (h - l + 1 <= max_range) {                                                         \
            uint64_t c_ = orc_markers_at(x, l, h, NULL);                                      \
            if (c_) {                                                                         \
                if (total + c_ > acc_cap) { acc_cap = (total + c_) * 2; acc = (uint64_t *)realloc(acc, acc_cap * 8); } \
                memmove(acc + c_, acc, total * 8);                                            \
                orc_markers_at(x, l, h, acc);                                                 \
                total += c_;                                                                  \
            }                                                                                 \
        }                                                                                     \
    } while (0)
    for (uint64_t i = 0; i < m; ++i) {
        orc_LF(x, l, h, q[m - i - 1], &l, &h);
        if (h < l) { free(acc); *lo = 1; *hi = 0; return 0; }
        if (window_ei - (m - i) >= wsize) {
            PREPEND_QUERY();
            window_ei = m - i;
        }
    }
    if (h >= l && (m - 1) % wsize != 0) PREPEND_QUERY();
#undef PREPEND_QUERY
    *lo = l; *hi = h;
    if (out) for (uint64_t i = 0; i < total && i < cap; ++i) out[i] = acc[i];
    free(acc);
    return total;
}

/* search_ftab (rowbowt.hpp:746-758) for the k-mer q[e-K, e) when the ftab is the one build_ftab(K)
 * makes for this index (rowbowt.hpp:726-744): its keys are the k-mers over ACGT with a non-empty
 * range and its values are find_range's, so a lookup is find_range on an ACGT-only k-mer. */
static int ftab_hit(const orc_index *x, const uint8_t *q, uint64_t e, uint64_t K, uint64_t *l, uint64_t *h) {
    for (uint64_t t = e - K; t < e; ++t)
        if (q[t] != 'A' && q[t] != 'C' && q[t] != 'G' && q[t] != 'T') return 0;
    orc_find_range(x, q + (e - K), K, l, h);
    return *h >= *l;
}

/* rowbowt.hpp:406-482.  K == 0: ft_ == nullptr (rb_markers' default).  K > 0: with the ftab of
 * k-mer size K loaded (rb_markers --ftab): the first K bases and every restart after a failed seed
 * go through search_ftab (:430-433, :454-464; a k-mer absent from the table restarts from the full
 * range K bases further left, as the reference's loop does).  mbuf is appended to by every update_mbuf (markers_at
 * does not clear, :271-285, :437-441) and cleared only after a failed seed was reported (:449).
 * A read shorter than K makes the reference throw (substr, :431); here it is treated as a miss. */
uint64_t orc_markers_greedy_seeding_ftab(const orc_index *x, const uint8_t *q, uint64_t m, uint64_t wsize, uint64_t max_range,
                                         uint64_t K, uint64_t *seeds, uint64_t cap_seeds, uint64_t *mk_out, uint64_t cap_mk,
                                         uint64_t *nmk) {
    const uint64_t fl = 0, fh = x->n - 1;          /* full_range(), :115-118 */
    uint64_t l = fl, h = fh, pl = fl, ph = fh;     /* range, prev_range :427-428 */
    uint64_t ns = 0, tot = 0, mb_begin = 0;        /* mbuf == markers [mb_begin, tot) */
#define UPDATE_MBUF(L_, H_)                                                                \
    do {                                                                                   \
        if ((H_) - (L_) + 1 <= max_range) {                                                \
            uint64_t c_ = orc_markers_at(x, (L_), (H_), NULL);                             \
            if (c_ && mk_out && tot + c_ <= cap_mk) orc_markers_at(x, (L_), (H_), mk_out + tot); \
            tot += c_;                                                                     \
        }                                                                                  \
    } while (0)
#define EMIT(L_, H_, QS_, QE_)                                                             \
    do {                                                                                   \
        if (seeds && ns < cap_seeds) {                                                     \
            uint64_t *d_ = seeds + 6 * ns;                                                 \
            d_[0] = (L_); d_[1] = (H_); d_[2] = (QS_); d_[3] = (QE_); d_[4] = mb_begin; d_[5] = tot; \
        }                                                                                  \
        ++ns;                                                                              \
    } while (0)
    uint64_t i = 0;
    if (K && m >= K) {                             /* :430-433 */
        uint64_t tl, th;
        if (ftab_hit(x, q, m, K, &tl, &th)) { l = tl; h = th; i = K; }
        pl = l; ph = h;
    }
    uint64_t window_ei = m, seed_ei = m;           /* :434 */
    for (; i < m; ++i) {
        orc_LF(x, l, h, q[m - i - 1], &l, &h);     /* :443 */
        if (h < l) {                               /* :444 the seed fails */
            if (seed_ei - (m - i) >= wsize) UPDATE_MBUF(pl, ph);   /* :445-447 */
            EMIT(pl, ph, m - i, seed_ei);          /* :448 fn(prev_range, (m-i, seed_ei-1), mbuf) */
            mb_begin = tot;                        /* :449 mbuf.clear() */
            pl = fl; ph = fh;                      /* :450 */
            seed_ei = m - i - 1;                   /* :452-453 */
            window_ei = m - i - 1;
            if (K && m - i - 1 >= K) {
                /* :454-464.  search_ftab answers a k-mer that is NOT in the table with {full_range(), 0}
                 * (:757), so the test "range.first <= range.second" (:459) holds for a miss as well: the
                 * loop always leaves on its first iteration -- a hit continues from the k-mer's range, a
                 * miss continues from the FULL range with the K bases skipped (i += K either way). */
                uint64_t tl, th;
                if (ftab_hit(x, q, m - i - 1, K, &tl, &th)) { l = tl; h = th; }
                else { l = fl; h = fh; }
                i += K;                            /* :460, then the outer ++i */
                pl = l; ph = h;                    /* :461 */
            } else { l = fl; h = fh; }             /* :466 */
        } else {
            if (window_ei - (m - i - 1) >= wsize) {    /* :469-472 */
                UPDATE_MBUF(l, h);
                window_ei = m - i - 1;
            }
            pl = l; ph = h;                        /* :473 */
        }
    }
    if (h >= l && seed_ei - (m - i) >= wsize) UPDATE_MBUF(l, h);   /* :478-480 */
    EMIT(l, h, m - i, seed_ei);                    /* :481 */
#undef UPDATE_MBUF
#undef EMIT
    if (nmk) *nmk = tot;
    return ns;
}

uint64_t orc_markers_greedy_seeding(const orc_index *x, const uint8_t *q, uint64_t m, uint64_t wsize, uint64_t max_range,
                                    uint64_t *seeds, uint64_t cap_seeds, uint64_t *mk_out, uint64_t cap_mk, uint64_t *nmk) {
    return orc_markers_greedy_seeding_ftab(x, q, m, wsize, max_range, 0, seeds, cap_seeds, mk_out, cap_mk, nmk);
}

/* rowbowt.hpp:222-256 + :664-685 */
uint64_t orc_greedy_locate(const orc_index *x, const uint8_t *q, uint64_t m, uint64_t min_length, uint64_t max_hits,
                           uint64_t *out, uint64_t cap,
                           uint64_t *seed_lo, uint64_t *seed_hi, uint64_t *seed_qs, uint64_t *seed_qe, uint64_t *seed_k) {
    if (!x->has_tsa) return 0;
    typedef struct { uint64_t lo, hi, qs, qe, k; } lfd_t;
    lfd_t *lfs = (lfd_t *)malloc((m + 2) * sizeof(lfd_t));
    uint64_t nlf = 0;
    uint64_t l = 0, h = x->n - 1, pl = 0, ph = x->n - 1;
    const uint64_t first_k = orc_last_run_sample(x);
    uint64_t k = first_k, pk = (uint64_t)-1, ei = m;
    for (uint64_t i = 0; i < m; ++i) {
        LF_w_loc(x, l, h, q[m - i - 1], k, &l, &h, &k);
        if (h < l) {
            if (ei - (m - i) >= min_length) { lfd_t d = { pl, ph, m - i, ei, pk }; lfs[nlf++] = d; }
            k = first_k;
            l = 0; h = x->n - 1; pl = 0; ph = x->n - 1;
            ei = m - i - 1;
        } else {
            pl = l; ph = h; pk = k;
        }
    }
    if (ei >= min_length) { lfd_t d = { pl, ph, 0, ei, pk }; lfs[nlf++] = d; }
    uint64_t cnt = 0;
    if (nlf) {
        lfd_t best = { 1, 0, 0, 0, 0 };
        uint64_t max_length = 0;
        for (uint64_t t = 0; t < nlf; ++t) {
            uint64_t length = lfs[t].qe - lfs[t].qs;
            if (length > max_length) { max_length = length; best = lfs[t]; }
        }
        uint64_t occ = best.hi >= best.lo ? best.hi - best.lo + 1 : 0;
        if (occ > max_hits) occ = max_hits;
        uint64_t *tmp = (uint64_t *)malloc((occ + 1) * 8);
        cnt = orc_locs_at(x, best.lo, best.hi, best.k, max_hits, tmp);
        for (uint64_t i = 0; i < cnt && i < cap; ++i) out[i] = tmp[i] - best.qs;
        free(tmp);
        if (seed_lo) *seed_lo = best.lo;
        if (seed_hi) *seed_hi = best.hi;
        if (seed_qs) *seed_qs = best.qs;
        if (seed_qe) *seed_qe = best.qe;
        if (seed_k) *seed_k = best.k;
    }
    free(lfs);
    return cnt;
}

/* doclist.hpp:46-50,77-79: rank over the doc-start bit-vector, capped at its size.
 * The bit-vector is sized from the LAST start read (doclist.hpp:66) and holds the starts in
 * sorted order; names stay in file order (doclist.hpp:49,72). */
static void docs_finish(orc_index *x) {
    x->doc_sorted = (uint64_t *)malloc((x->ndocs + 1) * 8);
    memcpy(x->doc_sorted, x->doc_starts, x->ndocs * 8);
    for (uint64_t a = 1; a < x->ndocs; ++a) { /* insertion sort: doc lists are tiny */
        uint64_t v = x->doc_sorted[a], b = a;
        while (b > 0 && x->doc_sorted[b - 1] > v) { x->doc_sorted[b] = x->doc_sorted[b - 1]; --b; }
        x->doc_sorted[b] = v;
    }
}
const char *orc_resolve_offset(const orc_index *x, uint64_t i, uint64_t *off) {
    *off = 0;
    if (!x->has_dl || x->ndocs == 0) return NULL;
    uint64_t size = x->doc_starts[x->ndocs - 1] + 1;
    uint64_t q = (i + 1 > size) ? size : i + 1;
    uint64_t rank = 0; /* # starts in [0,q) */
    while (rank < x->ndocs && x->doc_sorted[rank] < q) rank++;
    if (rank == 0) return NULL; /* reference indexes doc_names_[-1] here */
    *off = i - x->doc_sorted[rank - 1];
    return x->doc_names[rank - 1];
}

/* ------------------------------------------------------------------------------------------ */
/* loading                                                                                     */
/* ------------------------------------------------------------------------------------------ */
static int load_rbwt(orc_index *x, const char *fname) {
    rd_t r;
    if (rd_open(&r, fname)) { fprintf(stderr, "oracle: cannot open %s\n", fname); return -1; }
    /* rle_string.hpp:265-275 */
    x->n = rd_u64(&r);
    x->R = rd_u64(&r);
    x->B = rd_u64(&r);
    if (x->n != 0) {
        sdv_read(&r, &x->runs);
        for (int c = 0; c < 256; ++c) sdv_read(&r, &x->rpl[c]);
        hs_read(&r, &x->run_heads);
    }
    int bad = r.err || r.pos != r.len || x->run_heads.size != x->R;
    rd_close(&r);
    if (bad) { fprintf(stderr, "oracle: %s: unrecognised sdsl layout\n", fname); return -1; }
    build_f(x);
    return 0;
}

static int load_tsa(orc_index *x, const char *fname) {
    rd_t r;
    if (rd_open(&r, fname)) { fprintf(stderr, "oracle: cannot open %s\n", fname); return -1; }
    /* toehold_sa.hpp:85-91 */
    x->tsa_r = rd_u64(&r);
    x->tsa_n = rd_u64(&r);
    sdv_read(&r, &x->pred);
    iv_t sl, p2r;
    iv_read(&r, &sl, 1);
    iv_read(&r, &p2r, 1);
    int bad = r.err || r.pos != r.len || sl.len != x->tsa_r || p2r.len != x->tsa_r || x->pred.m != x->tsa_r;
    if (!bad) {
        x->samples_last = (uint64_t *)malloc(x->tsa_r * 8);
        x->pred_to_run = (uint64_t *)malloc(x->tsa_r * 8);
        for (uint64_t i = 0; i < x->tsa_r; ++i) { x->samples_last[i] = iv_get(&sl, i); x->pred_to_run[i] = iv_get(&p2r, i); }
        x->has_tsa = 1;
    }
    iv_free(&sl); iv_free(&p2r);
    rd_close(&r);
    if (bad) { fprintf(stderr, "oracle: %s: unrecognised sdsl layout\n", fname); return -1; }
    return 0;
}

static int load_mab(orc_index *x, const char *fname) {
    rd_t r;
    if (rd_open(&r, fname)) { fprintf(stderr, "oracle: cannot open %s\n", fname); return -1; }
    sdv_t rs, re, ai;
    sdv_read_raw(&r, &rs);
    sdv_read_raw(&r, &re);
    sdv_read_raw(&r, &ai);
    uint64_t count = rd_u64(&r);
    int bad = r.err || r.pos + count * 8 + 4 != r.len || rs.m != re.m || ai.m != rs.m;
    if (!bad) {
        x->ma_nruns = rs.m;
        x->ma_nvals = count;
        x->ma_start = rs.ones; rs.ones = NULL;
        x->ma_end = re.ones; re.ones = NULL;
        x->ma_off = (uint64_t *)malloc((x->ma_nruns + 1) * 8);
        for (uint64_t j = 0; j < x->ma_nruns; ++j) x->ma_off[j] = ai.ones[j];
        x->ma_off[x->ma_nruns] = count;
        x->ma_vals = (uint64_t *)malloc(count ? count * 8 : 8);
        rd_bytes(&r, x->ma_vals, count * 8);
        rd_bytes(&r, &x->ma_wsize, 4);
        x->has_ma = 1;
    }
    sdv_free(&rs); sdv_free(&re); sdv_free(&ai);
    rd_close(&r);
    if (bad) { fprintf(stderr, "oracle: %s: unrecognised marker-array layout\n", fname); return -1; }
    return 0;
}

/* doclist.hpp:57-73 */
static int load_docs(orc_index *x, const char *fname) {
    FILE *f = fopen(fname, "r");
    if (!f) { fprintf(stderr, "oracle: cannot open %s\n", fname); return -1; }
    char name[4096];
    unsigned long long pos;
    uint64_t cap = 16;
    x->doc_names = (char **)malloc(cap * sizeof(char *));
    x->doc_starts = (uint64_t *)malloc(cap * 8);
    x->ndocs = 0;
    while (fscanf(f, "%4095s %llu", name, &pos) == 2) {
        if (x->ndocs == cap) { cap *= 2; x->doc_names = (char **)realloc(x->doc_names, cap * sizeof(char *)); x->doc_starts = (uint64_t *)realloc(x->doc_starts, cap * 8); }
        x->doc_names[x->ndocs] = strdup(name);
        x->doc_starts[x->ndocs] = pos;
        x->ndocs++;
    }
    fclose(f);
    docs_finish(x);
    x->has_dl = 1;
    return 0;
}

orc_index *orc_load(const char *prefix, int flags) {
    orc_index *x = (orc_index *)calloc(1, sizeof(orc_index));
    size_t L = strlen(prefix) + 16;
    char *fn = (char *)malloc(L);
    int rc;
    snprintf(fn, L, "%s.rbwt", prefix);            /* rowbowt_io.hpp:17 */
    rc = load_rbwt(x, fn);
    if (!rc && (flags & ORC_SA)) { snprintf(fn, L, "%s.tsa", prefix); rc = load_tsa(x, fn); }   /* :18 */
    if (!rc && (flags & ORC_MA)) { snprintf(fn, L, "%s.mab", prefix); rc = load_mab(x, fn); }   /* :19 */
    if (!rc && (flags & ORC_DL)) { snprintf(fn, L, "%s.docs", prefix); rc = load_docs(x, fn); } /* :20 */
    free(fn);
    if (rc) { orc_free(x); return NULL; }
    return x;
}

/* rle_string.hpp:44-97 at run granularity: a set bit at the last position of every run whose
 * index is B-1 mod B except the final run (:68,:78); runs_per_letter[c] has a bit at the last
 * position (in c-only space) of each c-run (:65,:72,:77).
 * toehold_sa.hpp:133-155: sample = y ? y-1 : n-1; build_phi :105-131. */
orc_index *orc_build_from_runs(const uint8_t *heads, const uint64_t *lens, uint64_t R, uint64_t B,
                               const uint64_t *ssa_y, const uint64_t *esa_y) {
    orc_index *x = (orc_index *)calloc(1, sizeof(orc_index));
    x->B = B; x->R = R;
    uint64_t n = 0, cnt[256], nr[256];
    memset(cnt, 0, sizeof(cnt)); memset(nr, 0, sizeof(nr));
    for (uint64_t i = 0; i < R; ++i) { n += lens[i]; cnt[heads[i]] += lens[i]; nr[heads[i]]++; }
    x->n = n;
    x->runs.u = n;
    x->runs.ones = (uint64_t *)malloc((R / B + 1) * 8);
    x->runs.m = 0;
    for (int c = 0; c < 256; ++c) {
        x->rpl[c].u = cnt[c];
        x->rpl[c].m = 0;
        x->rpl[c].ones = nr[c] ? (uint64_t *)malloc(nr[c] * 8) : NULL;
    }
    x->run_heads.size = R;
    x->run_heads.heads = (uint8_t *)malloc(R ? R : 1);
    uint64_t pos = 0, cpos[256];
    memset(cpos, 0, sizeof(cpos));
    for (uint64_t i = 0; i < R; ++i) {
        uint8_t c = heads[i];
        x->run_heads.heads[i] = c;
        pos += lens[i];
        cpos[c] += lens[i];
        x->rpl[c].ones[x->rpl[c].m++] = cpos[c] - 1;
        if (i % B == B - 1 && i != R - 1) x->runs.ones[x->runs.m++] = pos - 1;
    }
    hs_index(&x->run_heads);
    build_f(x);
    if (ssa_y && esa_y) {
        x->tsa_r = R; x->tsa_n = n;
        x->samples_last = (uint64_t *)malloc(R * 8);
        x->pred_to_run = (uint64_t *)malloc(R * 8);
        /* samples_first sorted by text position (toehold_sa.hpp:108) */
        typedef struct { uint64_t pos, run; } pr_t;
        pr_t *sf = (pr_t *)malloc(R * sizeof(pr_t));
        for (uint64_t i = 0; i < R; ++i) {
            sf[i].pos = ssa_y[i] ? ssa_y[i] - 1 : n - 1;
            sf[i].run = i;
            x->samples_last[i] = esa_y[i] ? esa_y[i] - 1 : n - 1;
        }
        /* simple bottom-up merge sort on pos (stable; positions are distinct) */
        pr_t *tmp = (pr_t *)malloc(R * sizeof(pr_t));
        for (uint64_t w = 1; w < R; w *= 2) {
            for (uint64_t s = 0; s < R; s += 2 * w) {
                uint64_t a = s, am = s + w < R ? s + w : R, b = am, bm = s + 2 * w < R ? s + 2 * w : R, o = s;
                while (a < am && b < bm) tmp[o++] = (sf[b].pos < sf[a].pos) ? sf[b++] : sf[a++];
                while (a < am) tmp[o++] = sf[a++];
                while (b < bm) tmp[o++] = sf[b++];
            }
            pr_t *t = sf; sf = tmp; tmp = t;
        }
        x->pred.u = n; x->pred.m = R;
        x->pred.ones = (uint64_t *)malloc(R * 8);
        for (uint64_t i = 0; i < R; ++i) { x->pred.ones[i] = sf[i].pos; x->pred_to_run[i] = sf[i].run; }
        free(sf); free(tmp);
        x->has_tsa = 1;
    }
    return x;
}

/* SURVEY 8d's optional "reference-shaped" CPU mode: on = answer every rank / select / access of the rle_string and of
 * ToeholdSA's predecessor vector from Elias-Fano vectors and a Huffman-shaped wavelet tree (the structures sdsl holds for the
 * reference) instead of from the decoded arrays; same results, the reference's memory behaviour.  off = the plain arrays. */
void orc_set_reference_shaped(orc_index *x, int on) {
    if (!x) return;
    if (on) {
        ef_build(&x->runs);
        for (int c = 0; c < 256; ++c) ef_build(&x->rpl[c]);
        if (x->has_tsa) ef_build(&x->pred);
        wt_build(&x->run_heads);
    } else {
        ef_free(&x->runs);
        for (int c = 0; c < 256; ++c) ef_free(&x->rpl[c]);
        ef_free(&x->pred);
        wt_free(&x->run_heads);
    }
}

int orc_set_markers(orc_index *x, const uint64_t *run_start, const uint64_t *run_end, uint64_t nruns,
                    const uint64_t *mk_off, const uint64_t *mk_vals) {
    x->ma_nruns = nruns;
    x->ma_nvals = mk_off[nruns];
    x->ma_start = (uint64_t *)malloc((nruns + 1) * 8);
    x->ma_end = (uint64_t *)malloc((nruns + 1) * 8);
    x->ma_off = (uint64_t *)malloc((nruns + 1) * 8);
    x->ma_vals = (uint64_t *)malloc((x->ma_nvals + 1) * 8);
    memcpy(x->ma_start, run_start, nruns * 8);
    memcpy(x->ma_end, run_end, nruns * 8);
    memcpy(x->ma_off, mk_off, (nruns + 1) * 8);
    memcpy(x->ma_vals, mk_vals, x->ma_nvals * 8);
    x->has_ma = 1;
    return 0;
}

int orc_set_docs(orc_index *x, const char *names_joined, const uint64_t *starts, uint64_t ndocs) {
    x->doc_names = (char **)malloc((ndocs + 1) * sizeof(char *));
    x->doc_starts = (uint64_t *)malloc((ndocs + 1) * 8);
    const char *p = names_joined;
    for (uint64_t d = 0; d < ndocs; ++d) {
        x->doc_names[d] = strdup(p);
        p += strlen(p) + 1;
        x->doc_starts[d] = starts[d];
    }
    x->ndocs = ndocs;
    docs_finish(x);
    x->has_dl = 1;
    return 0;
}

void orc_free(orc_index *x) {
    if (!x) return;
    sdv_free(&x->runs);
    for (int c = 0; c < 256; ++c) sdv_free(&x->rpl[c]);
    hs_free(&x->run_heads);
    sdv_free(&x->pred);
    free(x->samples_last); free(x->pred_to_run);
    free(x->ma_start); free(x->ma_end); free(x->ma_off); free(x->ma_vals);
    if (x->doc_names) for (uint64_t d = 0; d < x->ndocs; ++d) free(x->doc_names[d]);
    free(x->doc_names); free(x->doc_starts); free(x->doc_sorted);
    free(x);
}

uint64_t orc_n(const orc_index *x) { return x->n; }
uint64_t orc_r(const orc_index *x) { return x->R; }
int orc_has_tsa(const orc_index *x) { return x->has_tsa; }
int orc_has_markers(const orc_index *x) { return x->has_ma; }
void orc_get_f(const orc_index *x, uint64_t f_out[256]) { memcpy(f_out, x->f, 256 * 8); }

void orc_get_runs(const orc_index *x, uint8_t *heads_out, uint64_t *lens_out) {
    for (uint64_t i = 0; i < x->R; ++i) { heads_out[i] = hs_at(&x->run_heads, i); lens_out[i] = run_at(x, i); }
}
void orc_get_tsa(const orc_index *x, uint64_t *pred_pos, uint64_t *samples_last, uint64_t *pred_to_run) {
    for (uint64_t i = 0; i < x->tsa_r; ++i) { pred_pos[i] = x->pred.ones[i]; samples_last[i] = x->samples_last[i]; pred_to_run[i] = x->pred_to_run[i]; }
}
uint64_t orc_marker_nruns(const orc_index *x) { return x->ma_nruns; }
uint64_t orc_marker_nvals(const orc_index *x) { return x->ma_nvals; }
void orc_get_markers(const orc_index *x, uint64_t *run_start, uint64_t *run_end, uint64_t *mk_off, uint64_t *mk_vals) {
    memcpy(run_start, x->ma_start, x->ma_nruns * 8);
    memcpy(run_end, x->ma_end, x->ma_nruns * 8);
    memcpy(mk_off, x->ma_off, (x->ma_nruns + 1) * 8);
    memcpy(mk_vals, x->ma_vals, x->ma_nvals * 8);
}

/* ------------------------------------------------------------------------------------------ */
/* batched drivers                                                                            */
/* ------------------------------------------------------------------------------------------ */
void orc_find_range_batch(const orc_index *x, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                          uint64_t *lo, uint64_t *hi, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)N; ++i)
        orc_find_range(x, seqs + off[i], off[i + 1] - off[i], &lo[i], &hi[i]);
}

void orc_find_range_w_toehold_batch(const orc_index *x, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                                    uint64_t *lo, uint64_t *hi, uint64_t *ssamp, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)N; ++i)
        orc_find_range_w_toehold(x, seqs + off[i], off[i + 1] - off[i], &lo[i], &hi[i], &ssamp[i]);
}

void orc_locs_at_batch(const orc_index *x, const uint64_t *lo, const uint64_t *hi, const uint64_t *k, uint64_t N,
                       uint64_t max_hits, const uint64_t *loc_off, uint64_t *locs, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)N; ++i)
        orc_locs_at(x, lo[i], hi[i], k[i], max_hits, locs + loc_off[i]);
}
