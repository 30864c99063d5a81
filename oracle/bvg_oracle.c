/*
 * bvg_oracle.c — CPU restatement of the BVGraph decode path (see bvg_oracle.h header comment).
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * The structure deliberately follows the reference's lazy-iterator composition
 *   Merged(Masked(blocks, refList), Merged(Intervals, Residuals), d)      BVGraph.java:1062-1090
 * so that the corner-case semantics (dedup on equal heads, cap at d, -1 after exhaustion, lazy
 * residual reads, recursive random access) are the reference's, not a re-derivation.
 */
#define _GNU_SOURCE
#include "bvg_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <pthread.h>

#define NO_INTERVALS 0                 /* BVGraph.java:379 */
#define INITIAL_LIST_LEN 1024          /* BVGraph.java:367 region: INITIAL_SUCCESSOR_LIST_LENGTH */

/* ------------------------------------------------------------------------------------------ */
/* Bit stream: dsiutils InputBitStream semantics — MSB-first within each byte.     */
/* ------------------------------------------------------------------------------------------ */

void bvgo_bits_init(bvgo_bits* b, const uint8_t* p, uint64_t nbytes, uint64_t pos) {
    b->p = p; b->nbits = nbytes * 8; b->pos = pos; b->err = 0;
}

/* Next 64 bits at the cursor, zero-padded past the end (no cursor movement). */
static inline uint64_t peek64(const bvgo_bits* b) {
    uint64_t byte = b->pos >> 3, nbytes = b->nbits >> 3;
    unsigned sh = (unsigned)(b->pos & 7);
    uint64_t hi = 0; uint8_t nxt = 0;
    if (byte + 9 <= nbytes) {
        uint64_t raw; memcpy(&raw, b->p + byte, 8);
        hi = __builtin_bswap64(raw);
        nxt = b->p[byte + 8];
    } else {
        for (int i = 0; i < 8; i++) hi = (hi << 8) | (byte + i < nbytes ? b->p[byte + i] : 0);
        nxt = (byte + 8 < nbytes) ? b->p[byte + 8] : 0;
    }
    return sh ? (hi << sh) | ((uint64_t)nxt >> (8 - sh)) : hi;
}

static inline void skip_bits(bvgo_bits* b, uint64_t n) {
    b->pos += n;
    if (b->pos > b->nbits) b->err = BVGO_E_EOF;
}

/* InputBitStream.readLong(n)/readInt(n): n bits, MSB first. */
uint64_t bvgo_read_bits(bvgo_bits* b, int n) {
    if (n == 0) return 0;
    uint64_t w = peek64(b);
    skip_bits(b, (uint64_t)n);
    return n >= 64 ? w : (w >> (64 - n));
}

/* InputBitStream.readUnary(): number of zeros before the first one. */
uint64_t bvgo_read_unary(bvgo_bits* b) {
    uint64_t z = 0;
    for (;;) {
        uint64_t w = peek64(b);
        if (w) {
            int lz = __builtin_clzll(w);
            skip_bits(b, (uint64_t)lz + 1);
            return z + (uint64_t)lz;
        }
        if (b->pos + 64 >= b->nbits) { b->pos = b->nbits + 1; b->err = BVGO_E_EOF; return z; }
        b->pos += 64; z += 64;
    }
}

/* InputBitStream.readGamma()/readLongGamma(): unary(msb) then msb low bits of x+1. */
uint64_t bvgo_read_gamma(bvgo_bits* b) {
    uint64_t msb = bvgo_read_unary(b);
    if (msb > 63) { b->err = BVGO_E_EOF; return 0; }
    return (((uint64_t)1 << msb) | bvgo_read_bits(b, (int)msb)) - 1;
}

/* InputBitStream.readDelta(): gamma(msb) then msb low bits of x+1. */
uint64_t bvgo_read_delta(bvgo_bits* b) {
    uint64_t msb = bvgo_read_gamma(b);
    if (msb > 63) { b->err = BVGO_E_EOF; return 0; }
    return (((uint64_t)1 << msb) | bvgo_read_bits(b, (int)msb)) - 1;
}

/* InputBitStream.readZeta(k)/readLongZeta(k) (verified on the fixture for k=3). */
uint64_t bvgo_read_zeta(bvgo_bits* b, int k) {
    uint64_t h = bvgo_read_unary(b);
    if (h * (uint64_t)k + (uint64_t)k - 1 > 63) { b->err = BVGO_E_EOF; return 0; }
    uint64_t left = (uint64_t)1 << (h * (uint64_t)k);
    uint64_t m = bvgo_read_bits(b, (int)(h * (uint64_t)k + (uint64_t)k - 1));
    if (m < left) return m + left - 1;
    return (m << 1) + bvgo_read_bits(b, 1) - 1;
}

/* InputBitStream.readNibble(): groups of (stop flag, 3 payload bits), flag = 1 on the last group. UNPINNED. */
uint64_t bvgo_read_nibble(bvgo_bits* b) {
    uint64_t x = 0, stop;
    do {
        x <<= 3;
        stop = bvgo_read_bits(b, 1);
        x |= bvgo_read_bits(b, 3);
    } while (!stop && !b->err);
    return x;
}

/* InputBitStream.readGolomb(m): unary quotient, minimal-binary remainder. UNPINNED. */
uint64_t bvgo_read_golomb(bvgo_bits* b, uint64_t m) {
    if (m == 0) return 0;
    uint64_t q = bvgo_read_unary(b);
    if (m == 1) return q;
    int log2b = 63 - __builtin_clzll(m);
    uint64_t thr = ((uint64_t)1 << (log2b + 1)) - m;
    uint64_t x = bvgo_read_bits(b, log2b);
    if (x >= thr) x = ((x << 1) + bvgo_read_bits(b, 1)) - thr;
    return q * m + x;
}

/* Fast.nat2int: 0,-1,1,-2,2,... */
int64_t bvgo_nat2int(uint64_t u) {
    return (u & 1) ? -(int64_t)((u + 1) >> 1) : (int64_t)(u >> 1);
}

/* ------------------------------------------------------------------------------------------ */
/* Graph handle                                                                                */
/* ------------------------------------------------------------------------------------------ */

struct bvgo_graph {
    bvgo_params p;
    const uint8_t* graph; uint64_t nbytes;
    const uint64_t* offsets;          /* n+1 entries or NULL */
    uint8_t* own_graph; uint64_t* own_offsets;
    /* one-entry outdegree cache, BVGraph.java:844-851 */
    int64_t cached_node; int64_t cached_outdegree; uint64_t cached_pointer;
};

void bvgo_default_params(bvgo_params* p) {
    memset(p, 0, sizeof *p);
    p->window_size = 7; p->max_ref_count = 3; p->min_interval_length = 4; p->zeta_k = 3;   /* BVGraph.java:455-473 */
    p->outdegree_coding = BVGO_GAMMA; p->block_coding = BVGO_GAMMA; p->residual_coding = BVGO_ZETA;
    p->reference_coding = BVGO_UNARY; p->block_count_coding = BVGO_GAMMA; p->offset_coding = BVGO_GAMMA; /* :527-542 */
}

static int coding_from_name(const char* s, size_t n) {
    static const struct { const char* name; int id; } tab[] = {
        {"DELTA", BVGO_DELTA}, {"GAMMA", BVGO_GAMMA}, {"GOLOMB", BVGO_GOLOMB}, {"SKEWED_GOLOMB", BVGO_SKEWED_GOLOMB},
        {"UNARY", BVGO_UNARY}, {"ZETA", BVGO_ZETA}, {"NIBBLE", BVGO_NIBBLE}};
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++)
        if (strlen(tab[i].name) == n && !memcmp(tab[i].name, s, n)) return tab[i].id;
    return -1;
}

/* string2Flags, BVGraph.java:1316-1331: names are FIELD_CODING joined by '|'. */
static int apply_flag(bvgo_params* p, const char* s, size_t n) {
    static const struct { const char* prefix; int field; } f[] = {
        {"OUTDEGREES_", 0}, {"BLOCKS_", 1}, {"RESIDUALS_", 2}, {"REFERENCES_", 3}, {"BLOCK_COUNT_", 4}, {"OFFSETS_", 5}};
    /* BLOCK_COUNT_ must be tested before BLOCKS_? they differ at char 5 ('S' vs '_'), no ambiguity. */
    for (size_t i = 0; i < 6; i++) {
        size_t pl = strlen(f[i].prefix);
        if (n > pl && !memcmp(s, f[i].prefix, pl)) {
            int id = coding_from_name(s + pl, n - pl);
            if (id < 0) return BVGO_E_IO;
            /* only the constants BVGraph declares exist (BVGraph.java:476-524); anything else is "unknown" (:1326) */
            static const unsigned allowed[6] = {
                1u << BVGO_GAMMA | 1u << BVGO_DELTA,
                1u << BVGO_GAMMA | 1u << BVGO_DELTA,
                1u << BVGO_GAMMA | 1u << BVGO_ZETA | 1u << BVGO_DELTA | 1u << BVGO_NIBBLE | 1u << BVGO_GOLOMB,
                1u << BVGO_GAMMA | 1u << BVGO_DELTA | 1u << BVGO_UNARY,
                1u << BVGO_GAMMA | 1u << BVGO_DELTA | 1u << BVGO_UNARY,
                1u << BVGO_GAMMA | 1u << BVGO_DELTA};
            if (!(allowed[f[i].field] >> id & 1u)) return BVGO_E_IO;
            switch (f[i].field) {
                case 0: p->outdegree_coding = id; break;
                case 1: p->block_coding = id; break;
                case 2: p->residual_coding = id; break;
                case 3: p->reference_coding = id; break;
                case 4: p->block_count_coding = id; break;
                case 5: p->offset_coding = id; break;
            }
            return 0;
        }
    }
    return BVGO_E_IO;
}

int bvgo_parse_properties(const char* text, size_t len, bvgo_params* out) {
    bvgo_params p; bvgo_default_params(&p);
    int have_nodes = 0, have_class = 0, version = 0;
    p.arcs = -1;
    size_t i = 0;
    while (i < len) {
        size_t e = i; while (e < len && text[e] != '\n' && text[e] != '\r') e++;
        size_t a = i; while (a < e && isspace((unsigned char)text[a])) a++;
        if (a < e && text[a] != '#' && text[a] != '!') {
            size_t k = a; while (k < e && text[k] != '=' && text[k] != ':' && !isspace((unsigned char)text[k])) k++;
            size_t v = k; while (v < e && isspace((unsigned char)text[v])) v++;
            if (v < e && (text[v] == '=' || text[v] == ':')) v++;
            while (v < e && isspace((unsigned char)text[v])) v++;
            size_t ve = e; while (ve > v && isspace((unsigned char)text[ve - 1])) ve--;
            char key[64] = {0}, val[512] = {0};
            size_t kl = k - a < 63 ? k - a : 63, vl = ve - v < 511 ? ve - v : 511;
            memcpy(key, text + a, kl); memcpy(val, text + v, vl);
            if (!strcmp(key, "nodes")) { p.nodes = strtoll(val, NULL, 10); have_nodes = 1; }
            else if (!strcmp(key, "arcs")) p.arcs = strtoll(val, NULL, 10);
            else if (!strcmp(key, "windowsize")) p.window_size = (int32_t)strtol(val, NULL, 10);
            else if (!strcmp(key, "maxrefcount")) p.max_ref_count = (int32_t)strtol(val, NULL, 10);
            else if (!strcmp(key, "minintervallength")) p.min_interval_length = (int32_t)strtol(val, NULL, 10);
            else if (!strcmp(key, "zetak")) p.zeta_k = (int32_t)strtol(val, NULL, 10);
            else if (!strcmp(key, "version")) version = (int)strtol(val, NULL, 10);
            else if (!strcmp(key, "graphclass")) {
                /* BVGraph.java:1491 + ImmutableGraph.java:687-691: both class names load. */
                const char* v2 = val; if (!strncmp(v2, "class ", 6)) v2 += 6;
                if (strcmp(v2, "it.unimi.dsi.big.webgraph.BVGraph") && strcmp(v2, "it.unimi.dsi.webgraph.BVGraph")) return BVGO_E_IO;
                have_class = 1;
            } else if (!strcmp(key, "compressionflags")) {
                size_t s = 0, n = strlen(val);
                while (s < n) {
                    while (s < n && (val[s] == '|' || isspace((unsigned char)val[s]))) s++;
                    size_t t = s; while (t < n && val[t] != '|' && !isspace((unsigned char)val[t])) t++;
                    if (t > s) { int r = apply_flag(&p, val + s, t - s); if (r) return r; }
                    s = t;
                }
            }
        }
        i = e; while (i < len && (text[i] == '\n' || text[i] == '\r')) i++;
    }
    if (!have_nodes || !have_class) return BVGO_E_IO;
    if (version > 0) return BVGO_E_IO;                  /* BVGraph.java:1496-1497 */
    *out = p;
    return 0;
}

static int check_codings(const bvgo_params* p) {
    /* allowed sets: the switch statements at BVGraph.java:628-632,655-659,695-700,729-734,759-764,788-795 */
    int o = p->outdegree_coding, r = p->reference_coding, bc = p->block_count_coding, b = p->block_coding, s = p->residual_coding, f = p->offset_coding;
    if (o != BVGO_GAMMA && o != BVGO_DELTA) return BVGO_E_UNSUPPORTED;
    if (r != BVGO_UNARY && r != BVGO_GAMMA && r != BVGO_DELTA) return BVGO_E_UNSUPPORTED;
    if (bc != BVGO_UNARY && bc != BVGO_GAMMA && bc != BVGO_DELTA) return BVGO_E_UNSUPPORTED;
    if (b != BVGO_UNARY && b != BVGO_GAMMA && b != BVGO_DELTA) return BVGO_E_UNSUPPORTED;
    if (s != BVGO_GAMMA && s != BVGO_ZETA && s != BVGO_DELTA && s != BVGO_GOLOMB && s != BVGO_NIBBLE) return BVGO_E_UNSUPPORTED;
    if (f != BVGO_GAMMA && f != BVGO_DELTA) return BVGO_E_UNSUPPORTED;
    return 0;
}

int bvgo_decode_offsets(const uint8_t* obytes, size_t nbytes, int64_t nodes, int coding, uint64_t* out) {
    bvgo_bits b; bvgo_bits_init(&b, obytes, nbytes, 0);
    uint64_t off = 0;
    for (int64_t i = 0; i <= nodes; i++) {          /* hasNext is i <= n: BVGraph.java:885 */
        uint64_t d = coding == BVGO_DELTA ? bvgo_read_delta(&b) : bvgo_read_gamma(&b);   /* readOffset :627-633 */
        if (b.err) return BVGO_E_EOF;
        out[i] = (off += d);
    }
    return 0;
}

static int read_file(const char* path, uint8_t** data, uint64_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) return BVGO_E_IO;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* d = (uint8_t*)malloc((size_t)sz + 16);
    if (!d) { fclose(f); return BVGO_E_NOMEM; }
    if (sz && fread(d, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(d); return BVGO_E_IO; }
    memset(d + sz, 0, 16);
    fclose(f);
    *data = d; *n = (uint64_t)sz;
    return 0;
}

int bvgo_open_mem(const bvgo_params* p, const uint8_t* graph, uint64_t nbytes, const uint64_t* offsets, bvgo_graph** out) {
    int r = check_codings(p); if (r) return r;
    bvgo_graph* g = (bvgo_graph*)calloc(1, sizeof *g);
    if (!g) return BVGO_E_NOMEM;
    g->p = *p; g->graph = graph; g->nbytes = nbytes; g->offsets = offsets;
    g->cached_node = -1;
    *out = g;
    return 0;
}

int bvgo_load(const char* basename, bvgo_graph** out) {
    char path[4096]; uint8_t *pt = NULL, *gb = NULL, *ob = NULL; uint64_t pn, gn, on;
    bvgo_params p;
    snprintf(path, sizeof path, "%s.properties", basename);
    int r = read_file(path, &pt, &pn); if (r) return r;
    r = bvgo_parse_properties((const char*)pt, pn, &p); free(pt); if (r) return r;
    snprintf(path, sizeof path, "%s.graph", basename);
    r = read_file(path, &gb, &gn); if (r) return r;
    uint64_t* offs = NULL;
    snprintf(path, sizeof path, "%s.offsets", basename);
    if (read_file(path, &ob, &on) == 0) {
        offs = (uint64_t*)malloc(((size_t)p.nodes + 1) * sizeof(uint64_t));
        if (!offs) { free(gb); free(ob); return BVGO_E_NOMEM; }
        r = bvgo_decode_offsets(ob, on, p.nodes, p.offset_coding, offs);
        free(ob);
        if (r) { free(gb); free(offs); return r; }
    }
    r = bvgo_open_mem(&p, gb, gn, offs, out);
    if (r) { free(gb); free(offs); return r; }
    (*out)->own_graph = gb; (*out)->own_offsets = offs;
    return 0;
}

void bvgo_close(bvgo_graph* g) { if (!g) return; free(g->own_graph); free(g->own_offsets); free(g); }
int bvgo_info(const bvgo_graph* g, bvgo_params* out) { *out = g->p; return 0; }
const uint64_t* bvgo_offsets(const bvgo_graph* g) { return g->offsets; }
const uint8_t* bvgo_graph_bytes(const bvgo_graph* g, uint64_t* nbytes) { if (nbytes) *nbytes = g->nbytes; return g->graph; }

/* ------------------------------------------------------------------------------------------ */
/* Field readers, BVGraph.java:654-796                                                         */
/* ------------------------------------------------------------------------------------------ */

static inline uint64_t read_coded(bvgo_bits* b, int coding, int k) {
    switch (coding) {
        case BVGO_GAMMA: return bvgo_read_gamma(b);
        case BVGO_DELTA: return bvgo_read_delta(b);
        case BVGO_UNARY: return bvgo_read_unary(b);
        case BVGO_ZETA: return bvgo_read_zeta(b, k);
        case BVGO_GOLOMB: return bvgo_read_golomb(b, (uint64_t)k);
        case BVGO_NIBBLE: return bvgo_read_nibble(b);
    }
    b->err = BVGO_E_UNSUPPORTED; return 0;
}
static inline int32_t read_outdegree(const bvgo_graph* g, bvgo_bits* b) { return (int32_t)read_coded(b, g->p.outdegree_coding, 0); }
static inline int64_t read_reference(const bvgo_graph* g, bvgo_bits* b) {
    uint64_t r = read_coded(b, g->p.reference_coding, 0);
    if (r > (uint64_t)g->p.window_size) { b->err = BVGO_E_STATE; return 0; }      /* BVGraph.java:701 */
    return (int64_t)r;
}
static inline int32_t read_block_count(const bvgo_graph* g, bvgo_bits* b) { return (int32_t)read_coded(b, g->p.block_count_coding, 0); }
static inline int32_t read_block(const bvgo_graph* g, bvgo_bits* b) { return (int32_t)read_coded(b, g->p.block_coding, 0); }
static inline uint64_t read_residual(const bvgo_graph* g, bvgo_bits* b) { return read_coded(b, g->p.residual_coding, g->p.zeta_k); }

/* outdegreeInternal, BVGraph.java:844-851 */
static int64_t outdegree_internal(bvgo_graph* g, int64_t x) {
    if (x == g->cached_node) return g->cached_outdegree;
    bvgo_bits b; bvgo_bits_init(&b, g->graph, g->nbytes, g->offsets[x]);
    int32_t d = read_outdegree(g, &b);
    if (b.err) return b.err;
    g->cached_node = x; g->cached_outdegree = d; g->cached_pointer = b.pos;
    return d;
}

int64_t bvgo_outdegree(bvgo_graph* g, int64_t x) {
    if (x == g->cached_node) return g->cached_outdegree;
    if (x < 0 || x >= g->p.nodes) return BVGO_E_ARG;            /* BVGraph.java:823 */
    if (!g->offsets) return BVGO_E_STATE;                        /* BVGraph.java:832 */
    return outdegree_internal(g, x);
}

/* ------------------------------------------------------------------------------------------ */
/* Lazy iterators                                                                              */
/* ------------------------------------------------------------------------------------------ */

typedef struct succ_it succ_it;

/* ResidualLongIterator, BVGraph.java:902-954 */
typedef struct { const bvgo_graph* g; bvgo_bits* ibs; int64_t next; int32_t remaining; } resid_it;
static void resid_init(resid_it* r, const bvgo_graph* g, bvgo_bits* ibs, int32_t count, int64_t x) {
    r->g = g; r->ibs = ibs; r->remaining = count;
    r->next = x + bvgo_nat2int(read_residual(g, ibs));                            /* :917 */
}
static inline int64_t resid_next(resid_it* r) {
    if (r->remaining == 0) return -1;
    int64_t result = r->next;
    if (--r->remaining != 0) r->next += (int64_t)read_residual(r->g, r->ibs) + 1;    /* :929 */
    return result;
}

/* LongIntervalSequenceIterator.java:57-78 */
typedef struct { int64_t* left; int64_t* len; int32_t remaining, curr_interval; int64_t curr_index, curr_left; } intv_it;
static void intv_init(intv_it* it, int64_t* left, int64_t* len, int32_t n) {
    it->left = left; it->len = len; it->remaining = n; it->curr_interval = 0; it->curr_index = 0;
    it->curr_left = n ? left[0] : 0;                                               /* :57-62 */
}
static inline int64_t intv_next(intv_it* it) {
    if (it->remaining == 0) return -1;                                            /* :72 */
    int64_t next = it->curr_left + it->curr_index++;
    if (it->curr_index == it->len[it->curr_interval]) {                           /* advance(), :64-68 */
        it->remaining--;
        if (it->remaining != 0) it->curr_left = it->left[++it->curr_interval];
        it->curr_index = 0;
    }
    return next;
}

/* LazyLongIterators.wrap(array, n), LazyLongIterators.java:220-255 */
typedef struct { const int64_t* a; int64_t n, i; } arr_it;

static int64_t succ_next(succ_it* s);
static void succ_free(succ_it* s);

/* underlying of a MaskedLongIterator: array (sequential mode) or recursive successors (random access) */
typedef struct { int is_arr; arr_it arr; succ_it* rec; } under_it;
static inline int64_t under_next(under_it* u) {
    if (u->is_arr) return u->arr.i < u->arr.n ? u->arr.a[u->arr.i++] : -1;
    return succ_next(u->rec);
}
static inline int64_t under_skip(under_it* u, int64_t n) {
    if (u->is_arr) { int64_t r = u->arr.n - u->arr.i; if (n > r) n = r; u->arr.i += n; return n; }
    int64_t i = 0;                                        /* AbstractLazyLongIterator.skip */
    while (i < n && succ_next(u->rec) != -1) i++;
    return i;
}

/* MaskedLongIterator.java:67-128 */
typedef struct { int64_t* mask; int32_t mask_len, curr_mask; int64_t left; under_it u; } masked_it;
static inline void masked_advance(masked_it* m) {
    if (m->left == 0 && m->curr_mask < m->mask_len) {
        under_skip(&m->u, m->mask[m->curr_mask++]);
        if (m->curr_mask < m->mask_len) m->left = m->mask[m->curr_mask++];
        else m->left = -1;
    }
}
static void masked_init(masked_it* m, int64_t* mask, int32_t mask_len) {
    m->mask = mask; m->mask_len = mask_len; m->curr_mask = 0;
    if (mask_len != 0) { m->left = mask[m->curr_mask++]; masked_advance(m); }
    else m->left = -1;
}
static inline int64_t masked_next(masked_it* m) {
    if (m->left == 0) return -1;
    int64_t next = under_next(&m->u);
    if (m->left == -1 || next == -1) return next;
    if (m->left > 0) { m->left--; masked_advance(m); }
    return next;
}

/* MergedLongIterator.java:54-92 semantics, generic over two "next" functions by macro-free struct. */
typedef struct { int64_t curr0, curr1, n; } merged_state;

struct succ_it {
    bvgo_bits* ibs;
    bvgo_bits* owned_bits;  /* heap stream of a recursive (random access) child, BVGraph.java:1084 */
    int kind;               /* 0 empty, 1 extras only, 2 blocks only, 3 merged(blocks, extras) */
    int extra_kind;         /* 0 none, 1 residuals, 2 intervals, 3 merged(intervals, residuals) */
    int64_t *block, *left, *len;
    masked_it masked; intv_it intv; resid_it resid;
    merged_state inner, outer;
    int err;
};

static inline int64_t extra_next(succ_it* s) {
    switch (s->extra_kind) {
        case 1: return resid_next(&s->resid);
        case 2: return intv_next(&s->intv);
        case 3: {
            merged_state* m = &s->inner;                                   /* MergedLongIterator.nextLong */
            if (m->n == 0 || (m->curr0 == -1 && m->curr1 == -1)) return -1;
            m->n--;
            int64_t result;
            if (m->curr0 == -1) { result = m->curr1; m->curr1 = resid_next(&s->resid); }
            else if (m->curr1 == -1) { result = m->curr0; m->curr0 = intv_next(&s->intv); }
            else if (m->curr0 < m->curr1) { result = m->curr0; m->curr0 = intv_next(&s->intv); }
            else if (m->curr0 > m->curr1) { result = m->curr1; m->curr1 = resid_next(&s->resid); }
            else { result = m->curr0; m->curr0 = intv_next(&s->intv); m->curr1 = resid_next(&s->resid); }
            return result;
        }
    }
    return -1;
}

static int64_t succ_next(succ_it* s) {
    switch (s->kind) {
        case 1: return extra_next(s);
        case 2: return masked_next(&s->masked);
        case 3: {
            merged_state* m = &s->outer;
            if (m->n == 0 || (m->curr0 == -1 && m->curr1 == -1)) return -1;
            m->n--;
            int64_t result;
            if (m->curr0 == -1) { result = m->curr1; m->curr1 = extra_next(s); }
            else if (m->curr1 == -1) { result = m->curr0; m->curr0 = masked_next(&s->masked); }
            else if (m->curr0 < m->curr1) { result = m->curr0; m->curr0 = masked_next(&s->masked); }
            else if (m->curr0 > m->curr1) { result = m->curr1; m->curr1 = extra_next(s); }
            else { result = m->curr0; m->curr0 = masked_next(&s->masked); m->curr1 = extra_next(s); }
            return result;
        }
    }
    return -1;
}

static void succ_release(succ_it* s) {
    if (!s->masked.u.is_arr && s->masked.u.rec) { succ_free(s->masked.u.rec); s->masked.u.rec = NULL; }
    free(s->block); free(s->left); free(s->len); free(s->owned_bits);
    s->block = s->left = s->len = NULL; s->owned_bits = NULL;
}
static void succ_free(succ_it* s) { if (!s) return; succ_release(s); free(s); }

/*
 * BVGraph.successors(x, ibs, window, outd), BVGraph.java:995-1097.
 * window == NULL  => random access (outdegreeInternal + recursion, :1006-1009, :1030, :1084)
 * window != NULL  => sequential (cyclic window of W+1 lists, :1010, :1018, :1081)
 * Returns d (>= 0) or a negative error; *s is initialised (caller releases with succ_release).
 */
static int64_t successors_setup(bvgo_graph* g, int64_t x, bvgo_bits* ibs, int64_t** window, int32_t* outd, succ_it* s, int depth) {
    memset(s, 0, sizeof *s);
    s->masked.u.is_arr = 1;
    s->ibs = ibs;
    if (x < 0 || x >= g->p.nodes) return BVGO_E_ARG;                              /* :1000 */
    const int W = g->p.window_size;
    const int64_t cyc = (int64_t)W + 1;                                           /* :1004 */
    int32_t d;
    if (!window) {
        int64_t dd = outdegree_internal(g, x); if (dd < 0) return dd;
        d = (int32_t)dd; ibs->pos = g->cached_pointer;                            /* :1007-1008 */
    } else {
        d = outd[x % cyc] = read_outdegree(g, ibs);                               /* :1010 */
        if (ibs->err) return ibs->err;
    }
    if (d == 0) { s->kind = 0; return 0; }                                        /* :1012 */
    int64_t ref = -1;
    if (W > 0) { ref = read_reference(g, ibs); if (ibs->err) return ibs->err; }   /* :1015-1016 */
    int64_t ref_index = (x - ref + cyc) % cyc;                                    /* :1018 */
    int32_t block_count = 0, extra_count;
    if (ref > 0) {                                                                /* :1020 */
        block_count = read_block_count(g, ibs); if (ibs->err) return ibs->err;
        if (block_count < 0) return BVGO_E_EOF;
        if (block_count) { s->block = (int64_t*)malloc(sizeof(int64_t) * (size_t)block_count); if (!s->block) return BVGO_E_NOMEM; }
        int32_t copied = 0, total = 0;
        for (int32_t i = 0; i < block_count; i++) {                               /* :1024-1028 */
            s->block[i] = (int64_t)read_block(g, ibs) + (i == 0 ? 0 : 1);
            if (ibs->err) return ibs->err;
            total += (int32_t)s->block[i];
            if ((i & 1) == 0) copied += (int32_t)s->block[i];
        }
        if ((block_count & 1) == 0) {                                             /* :1030 */
            int64_t dref;
            if (window) dref = outd[ref_index];
            else { dref = outdegree_internal(g, x - ref); if (dref < 0) return dref; }
            copied += (int32_t)dref - total;
        }
        extra_count = d - copied;                                                 /* :1031 */
    } else extra_count = d;                                                       /* :1033 */

    int32_t interval_count = 0;
    if (extra_count > 0) {                                                        /* :1037 */
        if (g->p.min_interval_length != NO_INTERVALS && (interval_count = (int32_t)bvgo_read_gamma(ibs)) != 0) {   /* :1040 */
            if (ibs->err) return ibs->err;
            if (interval_count < 0) return BVGO_E_EOF;
            s->left = (int64_t*)malloc(sizeof(int64_t) * (size_t)interval_count);
            s->len = (int64_t*)malloc(sizeof(int64_t) * (size_t)interval_count);
            if (!s->left || !s->len) return BVGO_E_NOMEM;
            int64_t prev;
            s->left[0] = prev = bvgo_nat2int(bvgo_read_gamma(ibs)) + x;           /* :1047 */
            s->len[0] = (int64_t)bvgo_read_gamma(ibs) + g->p.min_interval_length; /* :1048 */
            prev += s->len[0]; extra_count -= (int32_t)s->len[0];                 /* :1050-1051 */
            for (int32_t i = 1; i < interval_count; i++) {                        /* :1053-1058 */
                s->left[i] = prev = (int64_t)bvgo_read_gamma(ibs) + prev + 1;
                s->len[i] = (int64_t)bvgo_read_gamma(ibs) + g->p.min_interval_length;
                prev += s->len[i]; extra_count -= (int32_t)s->len[i];
                if (ibs->err) return ibs->err;
            }
        }
        if (ibs->err) return ibs->err;
    }
    const int32_t residual_count = extra_count;                                   /* :1062 */
    /* NB: Java's `residualCount == 0 ? null : new ResidualLongIterator(...)`; a negative count (malformed
       stream) would build an iterator that never stops at 0; we treat <= 0 as none. */
    int has_resid = residual_count > 0;
    if (has_resid) { resid_init(&s->resid, g, ibs, residual_count, x); if (ibs->err) return ibs->err; }   /* :1064 */
    if (interval_count) intv_init(&s->intv, s->left, s->len, interval_count);
    s->extra_kind = interval_count == 0 ? (has_resid ? 1 : 0) : (has_resid ? 3 : 2);                         /* :1067-1072 */
    if (s->extra_kind == 3) {                                                     /* MergedLongIterator ctor, n = Integer.MAX_VALUE (:45) */
        s->inner.n = INT32_MAX;
        s->inner.curr0 = intv_next(&s->intv);
        s->inner.curr1 = resid_next(&s->resid);
    }
    if (ref > 0) {                                                                /* :1074-1085 */
        if (window) {
            s->masked.u.is_arr = 1;
            s->masked.u.arr.a = window[ref_index]; s->masked.u.arr.n = outd[ref_index]; s->masked.u.arr.i = 0;   /* :1081 */
        } else {
            if (depth > 1 << 20) return BVGO_E_STATE;
            succ_it* rec = (succ_it*)malloc(sizeof *rec); if (!rec) return BVGO_E_NOMEM;
            bvgo_bits* rb = (bvgo_bits*)malloc(sizeof *rb); if (!rb) { free(rec); return BVGO_E_NOMEM; }
            bvgo_bits_init(rb, g->graph, g->nbytes, 0);                            /* new InputBitStream(graphMemory) */
            int64_t rd = successors_setup(g, x - ref, rb, NULL, NULL, rec, depth + 1);            /* :1084 */
            rec->owned_bits = rb;
            s->masked.u.is_arr = 0; s->masked.u.rec = rec;
            if (rd < 0) return rd;
        }
        masked_init(&s->masked, s->block, block_count);
    }
    if (ref <= 0) s->kind = s->extra_kind ? 1 : 0;                                /* :1087 */
    else if (s->extra_kind == 0) s->kind = 2;                                     /* :1088-1089 */
    else {
        s->kind = 3; s->outer.n = d;                                              /* :1090 */
        s->outer.curr0 = masked_next(&s->masked);
        s->outer.curr1 = extra_next(s);
    }
    return d;
}

int64_t bvgo_successors(bvgo_graph* g, int64_t x, int64_t* out, int64_t cap) {
    if (x < 0 || x >= g->p.nodes) return BVGO_E_ARG;                              /* :863 */
    if (!g->offsets) return BVGO_E_UNSUPPORTED;                                   /* :864 */
    bvgo_bits ibs; bvgo_bits_init(&ibs, g->graph, g->nbytes, 0);
    succ_it s;
    int64_t d = successors_setup(g, x, &ibs, NULL, NULL, &s, 0);
    if (d >= 0) {
        for (int64_t j = 0; j < d && j < cap; j++) out[j] = succ_next(&s);
        if (ibs.err) d = ibs.err;
    }
    succ_release(&s);
    return d;
}

/* ------------------------------------------------------------------------------------------ */
/* BVGraphNodeIterator, BVGraph.java:1100-1245                                                  */
/* ------------------------------------------------------------------------------------------ */

struct bvgo_iter {
    bvgo_graph* g;
    bvgo_bits ibs;
    int64_t cyc;
    int64_t** window; int64_t* wcap; int32_t* outd;
    int64_t from, curr, has_next_limit;
};

static int win_grow(bvgo_iter* it, int64_t idx, int64_t need) {
    if (it->wcap[idx] >= need) return 0;
    int64_t nc = it->wcap[idx] * 2; if (nc < need) nc = need;
    int64_t* p = (int64_t*)realloc(it->window[idx], sizeof(int64_t) * (size_t)nc);
    if (!p) return BVGO_E_NOMEM;
    it->window[idx] = p; it->wcap[idx] = nc;
    return 0;
}

void bvgo_iter_free(bvgo_iter* it) {
    if (!it) return;
    if (it->window) for (int64_t i = 0; i < it->cyc; i++) free(it->window[i]);
    free(it->window); free(it->wcap); free(it->outd); free(it);
}

int bvgo_node_iterator(bvgo_graph* g, int64_t from, bvgo_iter** out) {
    if (from < 0 || from > g->p.nodes) return BVGO_E_ARG;                         /* :1128 */
    bvgo_iter* it = (bvgo_iter*)calloc(1, sizeof *it);
    if (!it) return BVGO_E_NOMEM;
    it->g = g; it->cyc = (int64_t)g->p.window_size + 1; it->from = from;
    it->window = (int64_t**)calloc((size_t)it->cyc, sizeof(int64_t*));
    it->wcap = (int64_t*)calloc((size_t)it->cyc, sizeof(int64_t));
    it->outd = (int32_t*)calloc((size_t)it->cyc, sizeof(int32_t));
    if (!it->window || !it->wcap || !it->outd) { bvgo_iter_free(it); return BVGO_E_NOMEM; }
    for (int64_t i = 0; i < it->cyc; i++) {
        it->window[i] = (int64_t*)malloc(sizeof(int64_t) * INITIAL_LIST_LEN); it->wcap[i] = INITIAL_LIST_LEN;
        if (!it->window[i]) { bvgo_iter_free(it); return BVGO_E_NOMEM; }
    }
    bvgo_bits_init(&it->ibs, g->graph, g->nbytes, 0);
    if (from != 0) {                                                              /* :1135-1146 warm-up */
        if (!g->offsets) { bvgo_iter_free(it); return BVGO_E_STATE; }
        int64_t lim = from + 1 < it->cyc ? from + 1 : it->cyc;
        for (int64_t i = 1; i < lim; i++) {
            int64_t pos = (from - i + it->cyc) % it->cyc;
            int64_t d = outdegree_internal(g, from - i);
            if (d < 0) { bvgo_iter_free(it); return (int)d; }
            it->outd[pos] = (int32_t)d;
            int r = win_grow(it, pos, d); if (r) { bvgo_iter_free(it); return r; }
            int64_t dd = bvgo_successors(g, from - i, it->window[pos], d);
            if (dd < 0) { bvgo_iter_free(it); return (int)dd; }
        }
        it->ibs.pos = g->offsets[from];                                           /* :1144 */
    }
    it->curr = from - 1;                                                          /* :1147 */
    it->has_next_limit = g->p.nodes - 1;                                          /* :1148 with upperBound = Long.MAX_VALUE */
    *out = it;
    return 0;
}

void bvgo_iter_set_upper_bound(bvgo_iter* it, int64_t upper) {
    it->has_next_limit = (upper < it->g->p.nodes ? upper : it->g->p.nodes) - 1;   /* :1148 */
}
int bvgo_iter_has_next(const bvgo_iter* it) { return it->curr < it->has_next_limit; }   /* :1179-1181 */

int64_t bvgo_iter_next(bvgo_iter* it) {                                           /* :1164-1176 */
    if (!bvgo_iter_has_next(it)) return -1;
    int64_t x = ++it->curr;
    int64_t idx = x % it->cyc;
    succ_it s;
    int64_t d = successors_setup(it->g, x, &it->ibs, it->window, it->outd, &s, 0);
    if (d < 0) { succ_release(&s); return d < -1 ? d : BVGO_E_EOF; }
    int r = win_grow(it, idx, d); if (r) { succ_release(&s); return r; }
    /* NB: the referenced list may be window[idx] itself only if ref == 0 mod cyc, which ref<=W excludes. */
    int64_t* w = it->window[idx];
    for (int64_t j = 0; j < d; j++) w[j] = succ_next(&s);
    succ_release(&s);
    if (it->ibs.err) return it->ibs.err;
    return x;
}
int64_t bvgo_iter_outdegree(const bvgo_iter* it) { return it->curr == it->from - 1 ? BVGO_E_STATE : it->outd[it->curr % it->cyc]; }   /* :1206-1209 */
const int64_t* bvgo_iter_successors(const bvgo_iter* it) { return it->curr == it->from - 1 ? NULL : it->window[it->curr % it->cyc]; } /* :1192-1203 */
uint64_t bvgo_iter_bit_position(const bvgo_iter* it) { return it->ibs.pos; }

int bvgo_decode_range(bvgo_graph* g, int64_t from, int64_t to, int32_t* outdeg, int64_t* succ, uint64_t cap, uint64_t* n_succ) {
    if (from < 0 || to > g->p.nodes || from > to) return BVGO_E_ARG;
    bvgo_iter* it; int r = bvgo_node_iterator(g, from, &it); if (r) return r;
    bvgo_iter_set_upper_bound(it, to);
    uint64_t k = 0;
    while (bvgo_iter_has_next(it)) {
        int64_t x = bvgo_iter_next(it);
        if (x < 0) { bvgo_iter_free(it); return (int)x; }
        int64_t d = bvgo_iter_outdegree(it);
        if (outdeg) outdeg[x - from] = (int32_t)d;
        const int64_t* s = bvgo_iter_successors(it);
        for (int64_t j = 0; j < d; j++) { if (succ && k < cap) succ[k] = s[j]; k++; }
    }
    bvgo_iter_free(it);
    if (n_succ) *n_succ = k;
    return (succ && k > cap) ? BVGO_E_ARG : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Scan checksum (definition shared with include/bvgraph_hip.h)                                */
/* ------------------------------------------------------------------------------------------ */

static inline void node_key(uint64_t x, uint32_t* k0, uint32_t* k1) {           /* include/bvgraph_hip.h: the per-node key */
    uint32_t h = (uint32_t)x * 0x9E3779B1u + (uint32_t)(x >> 32) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    *k1 = h | 1u;
    uint32_t z = h * 0x297A2D39u; z ^= z >> 15; *k0 = z;
}
static inline uint64_t mix_keyed(uint32_t k0, uint32_t k1, uint64_t y) { return (uint64_t)k1 * y + (uint64_t)k0; }   /* mod 2^64 */
uint64_t bvgo_mix(uint64_t x, uint64_t y) {
    uint32_t k0, k1; node_key(x, &k0, &k1);
    return mix_keyed(k0, k1, y);
}

int bvgo_scan(bvgo_graph* g, int64_t from, int64_t to, uint64_t node_base, bvgo_scan_result* out) {
    if (from < 0 || to > g->p.nodes || from > to) return BVGO_E_ARG;
    bvgo_iter* it; int r = bvgo_node_iterator(g, from, &it); if (r) return r;
    bvgo_iter_set_upper_bound(it, to);
    uint64_t arcs = 0, chk = 0, nodes = 0;
    while (bvgo_iter_has_next(it)) {                                              /* SpeedTest.java:127-135 */
        int64_t x = bvgo_iter_next(it);
        if (x < 0) { bvgo_iter_free(it); return (int)x; }
        int64_t d = bvgo_iter_outdegree(it);
        const int64_t* s = bvgo_iter_successors(it);
        uint32_t k0, k1; node_key((uint64_t)x + node_base, &k0, &k1);
        for (int64_t j = 0; j < d; j++) chk += mix_keyed(k0, k1, (uint64_t)s[j] + node_base);
        arcs += (uint64_t)d; nodes++;
    }
    bvgo_iter_free(it);
    out->nodes = nodes; out->arcs = arcs; out->chk = chk;
    return 0;
}

typedef struct { bvgo_graph g; int64_t from, to; uint64_t base; bvgo_scan_result res; int rc; } mt_task;
static void* mt_run(void* a) { mt_task* t = (mt_task*)a; t->rc = bvgo_scan(&t->g, t->from, t->to, t->base, &t->res); return NULL; }

int bvgo_scan_mt(bvgo_graph* g, int64_t from, int64_t to, uint64_t node_base, int nthreads, bvgo_scan_result* out) {
    if (nthreads < 1) nthreads = 1;
    if (from < 0 || to > g->p.nodes || from > to) return BVGO_E_ARG;
    if (!g->offsets && nthreads > 1) return BVGO_E_STATE;
    mt_task* t = (mt_task*)calloc((size_t)nthreads, sizeof *t);
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof *th);
    if (!t || !th) { free(t); free(th); return BVGO_E_NOMEM; }
    int64_t n = to - from, m = (n + nthreads - 1) / nthreads;                     /* ImmutableGraph.java:415 */
    for (int i = 0; i < nthreads; i++) {
        t[i].g = *g; t[i].g.cached_node = -1; t[i].g.own_graph = NULL; t[i].g.own_offsets = NULL;   /* copy(): flyweight, BVGraph.java:553-578 */
        t[i].from = from + (int64_t)i * m; if (t[i].from > to) t[i].from = to;
        t[i].to = t[i].from + m; if (t[i].to > to) t[i].to = to;
        t[i].base = node_base;
        pthread_create(&th[i], NULL, mt_run, &t[i]);
    }
    int rc = 0; memset(out, 0, sizeof *out);
    for (int i = 0; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        if (t[i].rc && !rc) rc = t[i].rc;
        out->nodes += t[i].res.nodes; out->arcs += t[i].res.arcs; out->chk += t[i].res.chk;
    }
    free(t); free(th);
    return rc;
}

/* ---- arc labels stored as a bit stream (labelling/BitStreamArcLabelledImmutableGraph.java) ----
 * The node iterator reads `outdegree` labels at every nextLong() (:565-582), each with label.fromBitStream:
 * GammaCodedIntLabel.java:60-64 (readGamma) or FixedWidthIntLabel.java:70-73 (readInt(width)); random access positions the
 * stream at offset[x] (:208-229).  Parity: gamma is pinned through the cnr-2000 golden; the reference holds no labelled
 * fixture, so the layout itself (runs per node, gamma-coded run lengths in .labeloffsets) is restated from :75-84, :655-680. */
int bvgo_parse_label_spec(const char* spec, int* kind, int* width) {
    if (!spec || !kind || !width) return BVGO_E_ARG;
    const char* lp = strchr(spec, '('); const char* rp = strrchr(spec, ')');
    if (!lp || !rp || rp < lp) return BVGO_E_IO;
    const char* c0 = lp; while (c0 > spec && c0[-1] != '.') c0--;                 /* simple class name */
    size_t cl = (size_t)(lp - c0); while (cl && (c0[cl - 1] == ' ' || c0[cl - 1] == '\t')) cl--;
    if (cl == 18 && !strncmp(c0, "GammaCodedIntLabel", 18)) { *kind = BVGO_LABEL_GAMMA_INT; *width = 0; return 0; }
    if (cl == 18 && !strncmp(c0, "FixedWidthIntLabel", 18)) {
        const char* comma = memchr(lp, ',', (size_t)(rp - lp));
        if (!comma) return BVGO_E_IO;
        char* e = NULL; long w = strtol(comma + 1, &e, 10);
        if (e == comma + 1 || w < 0 || w > 32) return BVGO_E_IO;
        *kind = BVGO_LABEL_FIXED_INT; *width = (int)w; return 0;
    }
    if (cl == 22 && !strncmp(c0, "FixedWidthIntListLabel", 22)) {
        const char* comma = memchr(lp, ',', (size_t)(rp - lp));
        if (!comma) return BVGO_E_IO;
        char* e = NULL; long w = strtol(comma + 1, &e, 10);
        if (e == comma + 1 || w < 0 || w > 32) return BVGO_E_IO;
        *kind = BVGO_LABEL_FIXED_INT_LIST; *width = (int)w; return 0;
    }
    return BVGO_E_UNSUPPORTED;
}

/* FixedWidthIntListLabel.fromBitStream (FixedWidthIntListLabel.java:73-78): gamma(length), then `length` readInt(width).
 * list_off[arcs+1] = exclusive prefix of the lengths; values may be NULL to size it (cap = 0). */
int bvgo_labels_decode_lists(int width, const uint8_t* stream, uint64_t nbytes, const uint64_t* loffsets, int64_t nodes,
                             int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, int32_t* values, uint64_t cap, uint64_t* n_values) {
    if (from < 0 || to < from || to > nodes || !loffsets || !list_off) return BVGO_E_ARG;
    uint64_t a = 0, k = 0;
    list_off[0] = 0;
    for (int64_t x = from; x < to; x++) {
        bvgo_bits b; bvgo_bits_init(&b, stream, nbytes, loffsets[x]);
        for (int32_t j = 0; j < outdeg[x - from]; j++) {
            uint64_t len = bvgo_read_gamma(&b);
            if (b.err) return b.err;
            for (uint64_t t = 0; t < len; t++) {
                uint64_t v = bvgo_read_bits(&b, width);
                if (b.err) return b.err;
                if (values && k < cap) values[k] = (int32_t)(uint32_t)v;
                k++;
            }
            list_off[++a] = k;
        }
        if (b.pos != loffsets[x + 1]) return BVGO_E_EOF;
    }
    if (n_values) *n_values = k;
    return values && k > cap ? BVGO_E_ARG : 0;
}

/* FixedWidthLongListLabel.fromBitStream (labelling/FixedWidthLongListLabel.java:81-87): gamma(length), then `length` readLong(width),
 * width <= 64.  Test infrastructure like the rest of this file. */
int bvgo_labels_decode_lists64(int width, const uint8_t* stream, uint64_t nbytes, const uint64_t* loffsets, int64_t nodes,
                               int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, int64_t* values, uint64_t cap, uint64_t* n_values) {
    if (from < 0 || to < from || to > nodes || !loffsets || !list_off || width < 0 || width > 64) return BVGO_E_ARG;
    uint64_t a = 0, k = 0;
    list_off[0] = 0;
    for (int64_t x = from; x < to; x++) {
        bvgo_bits b; bvgo_bits_init(&b, stream, nbytes, loffsets[x]);
        for (int32_t j = 0; j < outdeg[x - from]; j++) {
            uint64_t len = bvgo_read_gamma(&b);
            if (b.err) return b.err;
            for (uint64_t t = 0; t < len; t++) {
                uint64_t v = 0;
                if (width > 32) { v = bvgo_read_bits(&b, width - 32) << 32; v |= bvgo_read_bits(&b, 32); } else v = bvgo_read_bits(&b, width);
                if (b.err) return b.err;
                if (values && k < cap) values[k] = (int64_t)v;
                k++;
            }
            list_off[++a] = k;
        }
        if (b.pos != loffsets[x + 1]) return BVGO_E_EOF;
    }
    if (n_values) *n_values = k;
    return values && k > cap ? BVGO_E_ARG : 0;
}

int bvgo_labels_decode(int kind, int width, const uint8_t* stream, uint64_t nbytes, const uint64_t* loffsets, int64_t nodes,
                       int64_t from, int64_t to, const int32_t* outdeg, int32_t* out, uint64_t cap, uint64_t* n_out) {
    if (from < 0 || to < from || to > nodes || !loffsets) return BVGO_E_ARG;
    if (kind != BVGO_LABEL_GAMMA_INT && kind != BVGO_LABEL_FIXED_INT) return BVGO_E_UNSUPPORTED;
    uint64_t total = 0;
    for (int64_t i = 0; i < to - from; i++) total += (uint64_t)outdeg[i];
    if (n_out) *n_out = total;
    if (total > cap) return BVGO_E_ARG;
    uint64_t k = 0;
    for (int64_t x = from; x < to; x++) {
        bvgo_bits b; bvgo_bits_init(&b, stream, nbytes, loffsets[x]);             /* ibs.position(offset.getLong(x)), :213 */
        for (int32_t j = 0; j < outdeg[x - from]; j++) {                           /* :579 */
            uint64_t v = kind == BVGO_LABEL_GAMMA_INT ? bvgo_read_gamma(&b) : bvgo_read_bits(&b, width);
            if (b.err) return b.err;
            out[k++] = (int32_t)(uint32_t)v;
        }
        if (b.pos != loffsets[x + 1]) return BVGO_E_EOF;                           /* the run must end where the next one starts */
    }
    return 0;
}

const char* bvgo_strerror(int code) {
    switch (code) {
        case 0: return "ok";
        case BVGO_E_ARG: return "node index out of range (IllegalArgumentException)";
        case BVGO_E_STATE: return "illegal state: reference > window or no offsets (IllegalStateException)";
        case BVGO_E_UNSUPPORTED: return "unsupported coding or access mode (UnsupportedOperationException)";
        case BVGO_E_IO: return "bad properties / missing file (IOException)";
        case BVGO_E_EOF: return "bit stream exhausted (EOFException)";
        case BVGO_E_NOMEM: return "out of memory";
    }
    return "unknown";
}
