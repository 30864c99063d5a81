/*
 * bvg_oracle.h — CPU restatement of the BVGraph successor-list decode path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link, load or call anything in oracle/.  The HIP
 * library (webgraph-big_amd/lib/libbvgraph_hip.so) never includes or links this file.
 *
 * What it restates (reference = vigna/webgraph-big 3.7.1, paths relative to /root/reference):
 *   src/it/unimi/dsi/big/webgraph/BVGraph.java:618-1265   decode half (read*, successors, node iterator)
 *   src/it/unimi/dsi/big/webgraph/BVGraph.java:1479-1574  loadInternal (properties, offsets)
 *   src/it/unimi/dsi/big/webgraph/MaskedLongIterator.java:67-128
 *   src/it/unimi/dsi/big/webgraph/MergedLongIterator.java:54-111
 *   src/it/unimi/dsi/big/webgraph/LongIntervalSequenceIterator.java:57-95
 *   src/it/unimi/dsi/big/webgraph/CompressionFlags.java:26-46
 * Third-party arithmetic that is NOT under /root/reference: dsiutils (it.unimi.dsi:dsiutils,
 * ivy.xml:20 rev="latest.release", i.e. unpinned) InputBitStream.{readUnary,readGamma,readDelta,
 * readZeta,readNibble,readGolomb} and Fast.nat2int — restated from their published algorithms.
 *
 * Parity pinning: gamma / unary / zeta_3 / nat2int are pinned end-to-end by the reference's own
 * golden fixture (test/.../BVGraphTest.java:105-123 testLarge: decode(cnr-2000.graph) must equal
 * cnr-2000.graph-txt.gz), which tests/test_oracle_golden.py replays on this oracle.
 * delta / nibble / Golomb / non-default codings: PARITY UNPINNED (the reference holds no vector).
 */
#ifndef BVG_ORACLE_H
#define BVG_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Coding ids, CompressionFlags.java:26-44 */
enum {
    BVGO_DELTA = 1, BVGO_GAMMA = 2, BVGO_GOLOMB = 3, BVGO_SKEWED_GOLOMB = 4,
    BVGO_UNARY = 5, BVGO_ZETA = 6, BVGO_NIBBLE = 7
};

/* Error codes (negative), mirroring the reference's exception classes. */
enum {
    BVGO_OK = 0,
    BVGO_E_ARG = -1,          /* IllegalArgumentException: node out of range */
    BVGO_E_STATE = -2,        /* IllegalStateException: ref > window, no offsets */
    BVGO_E_UNSUPPORTED = -3,  /* UnsupportedOperationException: coding not allowed for field */
    BVGO_E_IO = -4,           /* IOException: missing file, bad version / class / flag */
    BVGO_E_EOF = -5,          /* EOFException inside the bit stream */
    BVGO_E_NOMEM = -6
};

typedef struct {
    int64_t nodes;
    int64_t arcs;
    int32_t window_size;
    int32_t max_ref_count;
    int32_t min_interval_length;
    int32_t zeta_k;
    /* coding selectors, BVGraph.java:1281-1289 (defaults BVGraph.java:527-542) */
    int32_t outdegree_coding;
    int32_t block_coding;
    int32_t residual_coding;
    int32_t reference_coding;
    int32_t block_count_coding;
    int32_t offset_coding;
} bvgo_params;

typedef struct {
    uint64_t nodes;   /* nodes scanned */
    uint64_t arcs;    /* sum of outdegrees */
    uint64_t chk;     /* sum over arcs (x,y) of bvgo_mix(x,y) mod 2^64 */
} bvgo_scan_result;

typedef struct bvgo_graph bvgo_graph;
typedef struct bvgo_iter bvgo_iter;

/* Default parameters (BVGraph.java:455-473,527-542). */
void bvgo_default_params(bvgo_params* p);

/* Parses a Java .properties text (BVGraph.java:1479-1503, setFlags 1281-1331). */
int bvgo_parse_properties(const char* text, size_t len, bvgo_params* out);

/* Decodes the n+1 offset deltas (BVGraph.java:870-898, 1556-1558).  out has n+1 slots. */
int bvgo_decode_offsets(const uint8_t* obytes, size_t nbytes, int64_t nodes, int coding, uint64_t* out);

/* ImmutableGraph.load(basename): reads basename.{properties,graph,offsets}. */
int bvgo_load(const char* basename, bvgo_graph** out);
/* Same, from memory.  offsets may be NULL (sequential only, BVGraph.java:1136). Copies nothing:
 * the caller keeps graph/offsets alive for the lifetime of the handle. */
int bvgo_open_mem(const bvgo_params* p, const uint8_t* graph, uint64_t nbytes, const uint64_t* offsets, bvgo_graph** out);
void bvgo_close(bvgo_graph* g);
int bvgo_info(const bvgo_graph* g, bvgo_params* out);
const uint64_t* bvgo_offsets(const bvgo_graph* g);
const uint8_t* bvgo_graph_bytes(const bvgo_graph* g, uint64_t* nbytes);

/* BVGraph.outdegree(x), BVGraph.java:821-842.  Returns d >= 0 or a negative error. */
int64_t bvgo_outdegree(bvgo_graph* g, int64_t x);
/* BVGraph.successors(x) drained (random access incl. recursive reference chains, BVGraph.java:860-867,1084).
 * Writes min(d,cap) values (a -1 is written where the reference's iterator would return -1 early).
 * Returns d or a negative error. */
int64_t bvgo_successors(bvgo_graph* g, int64_t x, int64_t* out, int64_t cap);

/* BVGraph.nodeIterator(from) (BVGraph.java:1100-1265), incl. the warm-up of BVGraph.java:1135-1146. */
int bvgo_node_iterator(bvgo_graph* g, int64_t from, bvgo_iter** out);
/* NodeIterator.copy(upperBound) is modelled by bvgo_iter_set_upper_bound. */
void bvgo_iter_set_upper_bound(bvgo_iter* it, int64_t upper);
int bvgo_iter_has_next(const bvgo_iter* it);
/* nextLong(): returns the node, or -1 when exhausted (NoSuchElementException), or error < -1. */
int64_t bvgo_iter_next(bvgo_iter* it);
int64_t bvgo_iter_outdegree(const bvgo_iter* it);
/* successorBigArray(): pointer valid until the next bvgo_iter_next. */
const int64_t* bvgo_iter_successors(const bvgo_iter* it);
uint64_t bvgo_iter_bit_position(const bvgo_iter* it);
void bvgo_iter_free(bvgo_iter* it);

/* Full drain of [from,to) into caller arrays: outdeg[to-from], succ (cap elements).  Sequential path. */
int bvgo_decode_range(bvgo_graph* g, int64_t from, int64_t to, int32_t* outdeg, int64_t* succ, uint64_t cap, uint64_t* n_succ);

/* Scan checksum: the arc mix function shared (by definition, include/bvgraph_hip.h) with the HIP path. */
uint64_t bvgo_mix(uint64_t x, uint64_t y);
/* Sequential scan of [from,to): the SpeedTest loop (test/SpeedTest.java:127-135) plus checksum.
 * node_base is added to node ids and successors before mixing (shard of a larger graph). */
int bvgo_scan(bvgo_graph* g, int64_t from, int64_t to, uint64_t node_base, bvgo_scan_result* out);
/* Same split over nthreads contiguous node ranges as ImmutableGraph.splitNodeIterators (ImmutableGraph.java:405-436). */
int bvgo_scan_mt(bvgo_graph* g, int64_t from, int64_t to, uint64_t node_base, int nthreads, bvgo_scan_result* out);

/* ---- arc labels stored as a bit stream: labelling/BitStreamArcLabelledImmutableGraph.java:565-582 (sequential), :208-229
 * (random access); GammaCodedIntLabel.java:60-64, FixedWidthIntLabel.java:70-73.  loffsets: nodes+1 bit positions. ---- */
enum { BVGO_LABEL_GAMMA_INT = 1, BVGO_LABEL_FIXED_INT = 2, BVGO_LABEL_FIXED_INT_LIST = 3 };
int bvgo_parse_label_spec(const char* spec, int* kind, int* width);
int bvgo_labels_decode(int kind, int width, const uint8_t* stream, uint64_t nbytes, const uint64_t* loffsets, int64_t nodes,
                       int64_t from, int64_t to, const int32_t* outdeg, int32_t* out, uint64_t cap, uint64_t* n_out);

/* FixedWidthIntListLabel.java:73-78 */
int bvgo_labels_decode_lists(int width, const uint8_t* stream, uint64_t nbytes, const uint64_t* loffsets, int64_t nodes,
                             int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, int32_t* values, uint64_t cap, uint64_t* n_values);

const char* bvgo_strerror(int code);

/* ---- bit-level primitives exported for unit tests of the codes ---- */
typedef struct {
    const uint8_t* p;
    uint64_t nbits;   /* valid bits */
    uint64_t pos;     /* current bit */
    int err;          /* set to BVGO_E_EOF on over-read */
} bvgo_bits;
void bvgo_bits_init(bvgo_bits* b, const uint8_t* p, uint64_t nbytes, uint64_t pos);
uint64_t bvgo_read_bits(bvgo_bits* b, int n);
uint64_t bvgo_read_unary(bvgo_bits* b);
uint64_t bvgo_read_gamma(bvgo_bits* b);
uint64_t bvgo_read_delta(bvgo_bits* b);
uint64_t bvgo_read_zeta(bvgo_bits* b, int k);
uint64_t bvgo_read_nibble(bvgo_bits* b);
uint64_t bvgo_read_golomb(bvgo_bits* b, uint64_t m);
int64_t bvgo_nat2int(uint64_t u);

#ifdef __cplusplus
}
#endif
#endif
