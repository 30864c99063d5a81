/*
 * bvgraph_hip.h — C ABI of libbvgraph_hip.so: the MI355X (gfx950) BVGraph successor-list decoder.
 *
 * This is the drop-in boundary for ONE path of vigna/webgraph-big (reference paths relative to
 * /root/reference, BVG = src/it/unimi/dsi/big/webgraph/BVGraph.java, IG = .../ImmutableGraph.java):
 * the BVGraph decode half behind nodeIterator()/successors()/outdegree().  A JNI (or ctypes / C++)
 * shim binds exactly these entry points; INTEGRATION.md shows the Java side.
 *
 * Conventions: every function returns 0 or a negative bvg_status; no exceptions / longjmp cross the
 * boundary; all output buffers are caller-allocated host memory unless the name ends in _dev;
 * functions on ONE handle are not re-entrant, different handles (incl. bvg_copy() flyweights) are
 * (IG:187-197 threading contract).  There is no CPU fallback: without a gfx950 device every compute
 * entry point fails with BVG_E_HIP.
 */
#ifndef BVGRAPH_HIP_H
#define BVGRAPH_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVG_ABI_VERSION 3

/* Status codes; each maps 1:1 to the exception class the reference throws at the cited line. */
typedef enum bvg_status {
    BVG_OK = 0,
    BVG_E_ARG = -1,          /* IllegalArgumentException   BVG:823,863,1000,1128 (node out of range) */
    BVG_E_STATE = -2,        /* IllegalStateException      BVG:701 (ref > window), BVG:832,1136 (no offsets) */
    BVG_E_UNSUPPORTED = -3,  /* UnsupportedOperationException BVG:631,658,699,733,763,794,864 */
    BVG_E_IO = -4,           /* IOException                BVG:1492-1497,1326 (class / version / flag / file) */
    BVG_E_EOF = -5,          /* EOFException from the bit stream (record runs past the end of .graph) */
    BVG_E_NOMEM = -6,        /* OutOfMemoryError (host or device) */
    BVG_E_HIP = -7,          /* no gfx950 device / HIP runtime failure (no CPU fallback exists) */
    BVG_E_CAPACITY = -8      /* caller's successor buffer too small; *n_succ holds the size needed */
} bvg_status;

/* Coding ids, CompressionFlags.java:26-44. */
enum { BVG_DELTA = 1, BVG_GAMMA = 2, BVG_GOLOMB = 3, BVG_SKEWED_GOLOMB = 4, BVG_UNARY = 5, BVG_ZETA = 6, BVG_NIBBLE = 7 };

/* load_mode of bvg_open, = offsetType of BVG:1479 (loadInternal). */
enum { BVG_LOAD_OFFLINE = -1, BVG_LOAD_SEQUENTIAL = 0, BVG_LOAD_STANDARD = 1, BVG_LOAD_MAPPED = 2 };

/* The keys of basename.properties that drive decoding (BVG:1495-1503) + the six coding selectors
 * (setFlags, BVG:1281-1289; defaults BVG:527-542). */
typedef struct bvg_params {
    int64_t nodes;
    int64_t arcs;                 /* -1 if the properties do not say */
    int32_t window_size;          /* default 7  (BVG:455) */
    int32_t max_ref_count;        /* default 3  (BVG:461) */
    int32_t min_interval_length;  /* default 4  (BVG:467); 0 = NO_INTERVALS (BVG:379) */
    int32_t zeta_k;               /* default 3  (BVG:473) */
    int32_t outdegree_coding;     /* GAMMA | DELTA */
    int32_t block_coding;         /* GAMMA | DELTA | UNARY (the decoder's switch, BVG:759-764) */
    int32_t residual_coding;      /* ZETA | GAMMA | DELTA | GOLOMB | NIBBLE */
    int32_t reference_coding;     /* UNARY | GAMMA | DELTA */
    int32_t block_count_coding;   /* GAMMA | DELTA | UNARY */
    int32_t offset_coding;        /* GAMMA | DELTA */
} bvg_params;

/* Result of a fused on-chip scan (the SpeedTest loop, test/SpeedTest.java:127-141, plus a checksum). */
typedef struct bvg_scan_result {
    uint64_t nodes;        /* nodes scanned */
    uint64_t arcs;         /* sum of outdegrees */
    uint64_t chk;          /* sum over arcs (x,y) of bvg_arc_mix(x + node_base, y + node_base) mod 2^64 */
    uint64_t graph_bytes;  /* ALGORITHMIC bytes: compressed .graph bytes covering the scanned node range */
    uint64_t index_bytes;  /* device index bytes the kernels read on top (offsets + block plan + residual skip entries) */
    double kernel_ms;      /* hipEvent time of the scan kernel(s) on the handle's stream */
    uint32_t launches;     /* kernel launches issued (1 + slow-path relaunches) */
    uint32_t slow_blocks;  /* node blocks that had to take the global-memory slow path */
    uint64_t index_entries; /* residual skip entries of the scanned blocks that an index was present for (0 = the scan ran index-less) */
    uint32_t lean_blocks;  /* node blocks launched on the lean scan kernel (validated blocks of an indexed scan; bvg_scan.hip) */
    uint32_t reserved0;
} bvg_scan_result;

typedef struct bvg_graph bvg_graph;

/* ---- load (replaces ImmutableGraph.load -> BVGraph.loadInternal, IG:674-713, BVG:1479-1574) ---- */

void bvg_default_params(bvg_params* p);
/* Parses the text of a .properties file (BVG:1479-1503; class check BVG:1491, version BVG:1496). */
int bvg_parse_properties(const char* text, size_t len, bvg_params* out);
/* Decodes the n+1 gamma/delta coded offset gaps of basename.offsets (readOffset BVG:627-633,
 * OffsetsLongIterator BVG:870-898) into out[0..nodes]. Host-side, one-off at load (the reference
 * builds an Elias-Fano list here, BVG:1556-1558). */
int bvg_decode_offsets(const uint8_t* obytes, size_t nbytes, int64_t nodes, int coding, uint64_t* out);

/* BVGraph.load / loadMapped / loadOffline / loadSequential(basename) (BVG:1345-1464).  Reads
 * basename.properties/.graph[/.offsets], uploads to `device`.  BVG_LOAD_SEQUENTIAL / _OFFLINE need
 * no .offsets file: the index is then derived on the device (BVGraph -O / writeOffsets, BVG:2595-2609) by chunk-parallel speculative
 * walks iterated to the one consistent walk -- measured 0.4-10 s per GiB of stream on 0.25 GiB inputs, the fixed part being the
 * regions that settle one 4 KiB chunk per round (profiles/r03_derive_bench.txt); windows > 127 and streams the parallel walk finds
 * odd take one sequential pass of a single wavefront (~360 s per GiB). */
int bvg_open(const char* basename, int load_mode, int device, bvg_graph** out);
/* Same from host memory.  offsets: nodes+1 bit positions or NULL (derive on device). */
int bvg_open_mem(const bvg_params* p, const uint8_t* graph, uint64_t nbytes, const uint64_t* offsets, int device, bvg_graph** out);
/* Same from DEVICE memory already resident in HBM.  d_graph is adopted, not copied: it must stay
 * alive until bvg_close and be readable up to nbytes rounded up to 16 + 16 bytes.  d_offsets
 * (nodes+1 uint64) is read once: the library keeps its own packed index (4 bytes per node + 8 per
 * 1024 nodes, the counterpart of the Elias-Fano list of BVG:1545-1558) and the caller may free the
 * array when the call returns -- except when 1024 consecutive records span 2^32 bits or more (or
 * BVG_WIDE_OFFSETS=1 is set): then the plain array is used in place and must stay alive. */
int bvg_open_dev(const bvg_params* p, const void* d_graph, uint64_t nbytes, const void* d_offsets, int device, bvg_graph** out);
/* BVGraph.copy() (BVG:553-578): flyweight sharing the immutable device data, with its own stream
 * and workspace, usable from another thread. */
int bvg_copy(const bvg_graph* g, bvg_graph** out);
void bvg_close(bvg_graph* g);

int bvg_info(const bvg_graph* g, bvg_params* out);           /* numNodes/numArcs/windowSize/... */
/* The handle may stand for nodes [node_base, node_base + nodes) of a larger graph (a shard made by
 * ImmutableGraph.splitNodeIterators, IG:405-436): node ids and successors reported by scan /
 * decode are shifted by node_base.  Default 0. */
int bvg_set_node_base(bvg_graph* g, uint64_t node_base);
/* Copies the device offsets index back (nodes+1 entries): BVGraph -O / writeOffsets (BVG:2595-2609). */
int bvg_get_offsets(bvg_graph* g, uint64_t* out);

/* ---- decode (replaces BVG:821-867 outdegree/successors and BVG:1100-1245 BVGraphNodeIterator) ---- */

/* outdegree(x) for x in [from,to) (BVG:821-842). */
int bvg_outdegrees(bvg_graph* g, int64_t from, int64_t to, int32_t* out);
/* Materialises successors of nodes [from,to): outdeg[to-from] and the concatenated, strictly
 * increasing successor lists in succ (bit-exact with nodeIterator(from)...successorBigArray()).
 * to == from+1 is successors(x) (BVG:860-867).  If succ_cap is too small returns BVG_E_CAPACITY
 * with *n_succ = required size (succ may be NULL to query).  outdeg may be NULL. */
int bvg_decode_range(bvg_graph* g, int64_t from, int64_t to, int32_t* outdeg, int64_t* succ, uint64_t succ_cap, uint64_t* n_succ);
/* Same with the successors as 32-bit ids: graphs whose ids + node base stay BELOW 2^32 - 1 (nodes + node_base <= 0xFFFFFFFF; otherwise
 * BVG_E_UNSUPPORTED), host buffers only.  Half the bytes over PCIe, which bounds this path; the caller widens
 * (NodeIterator.successorBigArray() hands out longs, NodeIterator.java:80-96).  A missing successor of a malformed stream (-1 above) reads
 * 0xFFFFFFFF -- never a legal id here, which is why the limit is one below 2^32 -- and the widening caller maps it back to -1. */
int bvg_decode_range32(bvg_graph* g, int64_t from, int64_t to, int32_t* outdeg, uint32_t* succ, uint64_t succ_cap, uint64_t* n_succ);
/* Same, successor / outdegree buffers in device memory (stay in HBM for a downstream kernel). */
int bvg_decode_range_dev(bvg_graph* g, int64_t from, int64_t to, void* d_outdeg, void* d_succ, uint64_t succ_cap, uint64_t* n_succ);
/* Page-locked host memory for the buffers handed to bvg_decode_range / bvg_successors_batch: device -> host copies into it run
 * at the PCIe rate and need no staging (a JNI caller wraps it in a direct ByteBuffer, NewDirectByteBuffer).  Plain malloc'ed /
 * Java-heap buffers work too, only slower.  The iterator's buffer of NodeIterator.successorBigArray() (NodeIterator.java:80-96)
 * is the intended use: one pair of buffers per iterator, reused batch after batch. */
void* bvg_host_alloc(size_t bytes);
void bvg_host_free(void* p);
/* successors(x) for a whole frontier at once (BVG:860-867 per element; the access pattern of
 * algo/ParallelBreadthFirstVisit.java and algo/HyperBall.java:774-822): nodes[count] in any order, repeats
 * allowed; outdeg[count] and the successor lists concatenated in request order.  Each request is
 * decoded together with the few earlier nodes its reference chain reaches (the recursion of BVG:1084). */
int bvg_successors_batch(bvg_graph* g, const int64_t* nodes, int64_t count, int32_t* outdeg, int64_t* succ, uint64_t succ_cap, uint64_t* n_succ);
/* Full sequential successor scan of [from,to) consumed on-chip (arc count + checksum).
 * The first scan of >= 4096 nodes also builds the residual skip index of the node blocks it covers (a header walk that counts the
 * entries, a dense walk that fills them, and a validating decode that uses and checks them -- and reports this very scan's result, so
 * the first scan IS the build; a shard of a multi-GPU scan therefore indexes its own part only, a later scan of other nodes indexes the
 * whole graph); a materialising call (bvg_decode_range) builds it for the whole graph once it covers >= 1/4 of the nodes.  The index is shared by bvg_copy()
 * flyweights: 6 bytes (10 for graphs on the 64-bit successor kernels: more than 2^32 - 256 nodes) per 16 residuals of lists
 * with >= 16 residuals -- per 8 of lists with >= 8 on graphs with references below 40 arcs per node; the granularity is chosen when
 * the index is built and travels with it -- (cf. the offset cache the reference builds at load, BVG:1545-1558).  The same passes VALIDATE the
 * blocks: the lean scan kernel then skips the checks a well-formed stream cannot fail, blocks that failed one stay on the
 * checking kernels for good. */
int bvg_scan(bvg_graph* g, int64_t from, int64_t to, bvg_scan_result* out);
/* Builds that index for the blocks of [from,to) now (0, nodes = the whole graph) instead of inside the first scan; a no-op when
 * they are covered already.  entries / bytes (either may be NULL) report what the index of the graph holds afterwards. */
int bvg_build_index(bvg_graph* g, int64_t from, int64_t to, uint64_t* entries, uint64_t* bytes);
/* The device index on disk: the block plan and the residual skip index (with its validation marks) as they stand, written to `path` and
 * loaded back by a later process instead of rebuilt (cf. the reference's cached offsets big list, basename.obl, BVG:1545-1555).
 * bvg_open() loads basename.bvgidx by itself when it exists and is not older than basename.graph; a file that does not belong to
 * the graph (format 2: size, every parameter that shapes a record, block size, a hash of EVERY byte of the stream computed on the device) or is damaged (a
 * checksum over its payload; range checks on every array) is refused with BVG_E_IO and the index is built as usual: the lean kernels trust the marks. */
int bvg_save_index(bvg_graph* g, const char* path);
int bvg_load_index(bvg_graph* g, const char* path);
/* Node-range split points for k shards of ~equal compressed size (the balanced variant of
 * IG:405-436; cf. algo/HyperBall.java:748-768): bounds[0..k], bounds[0]=0, bounds[k]=nodes. */
int bvg_split_by_bits(bvg_graph* g, int k, int64_t* bounds);
/* The same with ~equal ARC counts per shard: bounds[j] = first node whose cumulative outdegree reaches j * arcs / k (the
 * skipTo() walk over algo/EliasFanoCumulativeOutdegreeList.java:30-75 that algo/HyperBall.java:748-768 uses for its tasks);
 * outdegrees and their prefix sum are computed on the device. */
int bvg_split_by_arcs(bvg_graph* g, int k, int64_t* bounds);

/* ---- node-range shards over several GPUs (ImmutableGraph.splitNodeIterators, IG:405-436; arc-balanced tasks as in
 * algo/HyperBall.java:748-768).  Shards are independent: each decodes its own node range (and re-derives its halo locally), the
 * only thing ever combined is {nodes, arcs, chk}, by a plain sum. ---- */
enum { BVG_BALANCE_NODES = 0,   /* ceil(n/k) nodes per shard: the reference's rule, IG:415-433 */
       BVG_BALANCE_BITS = 1,    /* ~equal compressed bits (bvg_split_by_bits) */
       BVG_BALANCE_ARCS = 2 };  /* ~equal arc counts (bvg_split_by_arcs) */
/* bounds[0..k] of the k-way split; cached in the graph, so flyweights and later calls agree. */
int bvg_shard_bounds(bvg_graph* g, int k, int balance, int64_t* bounds);
/* The scan of shard r of k: nodes [bounds[r], bounds[r+1]) (returned in *from / *to when not NULL).  One rank of a
 * one-process-per-GPU job calls this on its replica and all-reduces {arcs, chk} (RCCL: 16 bytes); the sum over r = 0..k-1
 * equals bvg_scan(g, 0, nodes). */
int bvg_scan_shard(bvg_graph* g, int k, int r, int balance, bvg_scan_result* out, int64_t* from, int64_t* to);
/* One process, ngpu devices (a JVM host): per_gpu[i] = a handle of the SAME graph on device i (or bvg_copy() flyweights on
 * one device); shard i runs on per_gpu[i], all shards concurrently, results summed on the host (total; per_shard[ngpu]
 * optional).  total->kernel_ms = the slowest shard. */
int bvg_scan_multi(bvg_graph* const* per_gpu, int ngpu, int balance, bvg_scan_result* total, bvg_scan_result* per_shard);

/* ---- transposition feed (the decode + sort of Transform.transposeOffline, Transform.java:1058-1160; processBatch :938) ----
 * Decodes every arc (x,y) of the graph on the device, sorts the pairs by target (stable radix sort, so sources stay increasing)
 * and returns the TRANSPOSE in CSR form: toffsets[nodes+1] = exclusive prefix of the in-degrees, tsucc[arcs] = for each node y
 * the sources of its incoming arcs in increasing order (what ArcListASCIIGraph / BVGraph.store of the transpose would list).
 * tsucc may be NULL to query *n_arcs (returns BVG_E_CAPACITY).  Requires node_base == 0.  _dev: both buffers in device memory. */
int bvg_transpose(bvg_graph* g, uint64_t* toffsets, int64_t* tsucc, uint64_t tsucc_cap, uint64_t* n_arcs);
int bvg_transpose_dev(bvg_graph* g, void* d_toffsets, void* d_tsucc, uint64_t tsucc_cap, uint64_t* n_arcs);

/* Transform.symmetrizeOffline (Transform.java:546-575) = union(g, transposeOffline(g)): the SYMMETRISED graph in CSR form —
 * soffsets[nodes+1], ssucc = for each node the increasing union of its successors and its predecessors (an arc present in both
 * directions once; loops kept).  *n_arcs = arcs of the result, known only after the transposition: a call with too small a
 * buffer (or ssucc NULL) returns BVG_E_CAPACITY with soffsets and *n_arcs filled and costs the full pass; 2 x numArcs() always
 * suffices.  Requires node_base == 0. */
int bvg_symmetrize(bvg_graph* g, uint64_t* soffsets, int64_t* ssucc, uint64_t ssucc_cap, uint64_t* n_arcs);
int bvg_symmetrize_dev(bvg_graph* g, void* d_soffsets, void* d_ssucc, uint64_t ssucc_cap, uint64_t* n_arcs);

/* ---- arc labels stored as a bit stream (labelling/BitStreamArcLabelledImmutableGraph.java) ----
 * basename.labels holds, node after node, the labels of the node's arcs in successor order (:75-84); basename.labeloffsets the
 * gamma-coded bit lengths of those runs after a leading gamma(0) (store(), :655-680).  The node iterator reads `outdegree`
 * labels per node (:565-582).  Built for the scalar label classes: GammaCodedIntLabel (GammaCodedIntLabel.java:60-64) and
 * FixedWidthIntLabel (FixedWidthIntLabel.java:70-73), and for FixedWidthIntListLabel (FixedWidthIntListLabel.java:73-78:
 * gamma length + elements of `width` bits per arc) and FixedWidthLongListLabel (elements of up to 64 bits, below); user label
 * classes return BVG_E_UNSUPPORTED. */
enum { BVG_LABEL_GAMMA_INT = 1, BVG_LABEL_FIXED_INT = 2, BVG_LABEL_FIXED_INT_LIST = 3, BVG_LABEL_FIXED_LONG_LIST = 4 };
typedef struct bvg_labels bvg_labels;
/* Label.toSpec() text, e.g. "it.unimi.dsi.big.webgraph.labelling.FixedWidthIntLabel(FOO,10)" -> kind, width. */
int bvg_labels_parse_spec(const char* spec, int* kind, int* width);
/* label_offsets: nodes+1 bit positions into the label stream (decode basename.labeloffsets with bvg_decode_offsets(.., BVG_GAMMA, ..)). */
int bvg_labels_open_mem(int kind, int width, int64_t nodes, const uint8_t* stream, uint64_t nbytes, const uint64_t* label_offsets, int device, bvg_labels** out);
/* basename.properties alone (host-only): label class and the basename of the underlying graph (property underlyinggraph,
 * resolved against the property file, :95-97). */
int bvg_labels_read_properties(const char* basename, int* kind, int* width, char* underlying, size_t underlying_cap);
/* BitStreamArcLabelledImmutableGraph.load (:378-484): reads basename.{properties,labels,labeloffsets}; `underlying` receives the
 * basename of the underlying graph (property underlyinggraph, resolved against the property file), to be opened with bvg_open.
 * nodes = numNodes() of that graph. */
int bvg_labels_open(const char* basename, int64_t nodes, int device, bvg_labels** out, char* underlying, size_t underlying_cap);
void bvg_labels_close(bvg_labels* l);
int bvg_labels_info(const bvg_labels* l, int* kind, int* width, int64_t* nodes, uint64_t* stream_bytes);
/* Labels of the arcs of nodes [from,to) in the order bvg_decode_range lists the successors; outdeg[to-from] as returned by it.
 * *n_labels = sum of the outdegrees; BVG_E_CAPACITY if cap is smaller; BVG_E_EOF if a node's run does not end at the next offset
 * (the outdegrees do not belong to this label stream). */
int bvg_labels_decode_range(bvg_labels* l, int64_t from, int64_t to, const int32_t* outdeg, int32_t* labels, uint64_t cap, uint64_t* n_labels);
/* List labels (kind BVG_LABEL_FIXED_INT_LIST): list_off[arcs+1] = exclusive prefix of the list lengths of the arcs of [from,to) in
 * successor order, values[cap] = the concatenated elements; *n_values = their number.  BVG_E_CAPACITY if cap is smaller (list_off is
 * filled either way: size the buffer from list_off[arcs] and call again). */
int bvg_labels_decode_range_lists(bvg_labels* l, int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, int32_t* values, uint64_t cap, uint64_t* n_values);
/* The same for FixedWidthLongListLabel (labelling/FixedWidthLongListLabel.java:81-87: gamma(length), then readLong(width), width <= 64):
 * kind BVG_LABEL_FIXED_LONG_LIST, 64-bit elements. */
int bvg_labels_decode_range_lists64(bvg_labels* l, int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, int64_t* values, uint64_t cap, uint64_t* n_values);
/* Same as bvg_labels_decode_range, outdegrees (int32) and labels (int32) in device memory: chains with bvg_decode_range_dev without leaving HBM. */
int bvg_labels_decode_range_dev(bvg_labels* l, int64_t from, int64_t to, const void* d_outdeg, void* d_labels, uint64_t cap, uint64_t* n_labels);

/* ---- the compressor on the device (second half): BVGraph.store (BVG:2329-2470; CompressionThread.call
 * :2216-2327, diffComp :1977-2159, intervalize :1595-1618) from an adjacency in CSR form -- adj_off[nodes+1], adj[adj_off[nodes]] with
 * strictly increasing successor lists -- to the bytes of basename.graph and the nodes+1 bit offsets (write basename.offsets from them
 * with the gamma / delta coded gaps of BVG:2228,2311).  p gives windowsize, maxrefcount (-1 = unbounded), minintervallength, zetak and
 * the codings (nodes / arcs are ignored).  chunk_nodes > 0 compresses ranges of that many nodes with a fresh window each, as the
 * reference's multi-threaded store does (BVG:2404-2457); 0 = one range = the single-threaded store, byte for byte.
 * *graph / *offsets are malloc'ed (bvg_free).  BVG_E_ARG for lists that are not strictly increasing or leave [0, nodes). */
int bvg_store(const bvg_params* p, int64_t nodes, const uint64_t* adj_off, const int64_t* adj, int64_t chunk_nodes, int device,
              uint8_t** graph, uint64_t* graph_bytes, uint64_t** offsets);
void bvg_free(void* p);

/* ---- synthetic-workload helper (bench only): K back-to-back copies of the graph ----
 * BV records are translation invariant (every value is coded relative to the node id, Appendix A.3
 * of SURVEY.md), so the concatenation of K copies of the bit stream is a valid BVGraph with K*nodes
 * nodes in which copy j is the base graph shifted by j*nodes.  Built on the device. */
int bvg_tile(const bvg_graph* base, int64_t copies, bvg_graph** out);
/* The same with k <= 16 different base graphs (same device, same BV parameters): the cycle {bases[0], ..., bases[k-1]} repeated `cycles`
 * times -- a synthetic workload whose tiles differ in seed and degree mix. */
int bvg_mosaic(const bvg_graph* const* bases, int k, int64_t cycles, bvg_graph** out);

/* ---- tuning knobs (optional) ---- */
typedef struct bvg_tuning {
    uint32_t block_bits;     /* target compressed bits per node block (one wavefront each); 0 = default */
    uint32_t force_wide;     /* 1 = use the 64-bit successor kernels even when every node id fits 32 bits (nodes <= 2^32 - 256); steady-state scans of
                                such a graph still run the scan kernel, on 32-bit lists of ids relative to a per-block base */
    uint32_t force_slow;     /* 1 = route every block through the global-memory slow path (tests) */
    uint32_t reserved;       /* low byte 2 = experimental streaming kernel as tier 0; bits 8.. = its grab threshold */
    uint32_t no_index;       /* 1 = calls on THIS handle neither build nor read the residual skip index (nor the validation marks, so every
                                block stays on the checking kernels): what a cold consumer gets from a graph it scans once; bench.py times
                                a bvg_copy() flyweight with it (`value_no_index`) beside the indexed steady state (ABI version 3).
                                2 = "marks only" (round 6): the index THIS handle builds holds the validation marks (one byte per block of
                                ~4 KiB of stream: the lean scan kernel takes the block) but entries only for lists of >= 4 096 residuals --
                                ~0.03 % of the stream instead of ~50 %; the residuals of a list are then one lane's walk.  Measured
                                (profiles/r06_ab_marks_*.txt): eu15 stand-in 113 G edges/s (indexed 310, checking kernels 45), cnr-2000
                                tiled 142 (151, 80).  An index that exists already is used as it is. */
} bvg_tuning;
int bvg_set_tuning(bvg_graph* g, const bvg_tuning* t);

const char* bvg_strerror(int status);
int bvg_abi_version(void);

/* ---- checksum definition (shared with the CPU oracle) ----
 *   h  = (u32)x * 0x9E3779B1 + (u32)(x >> 32) * 0x85EBCA77;  h ^= h >> 15;  h *= 0x2C1B3C6D;  h ^= h >> 12     (mod 2^32)
 *   k1 = h | 1;   k0 = h * 0x297A2D39;  k0 ^= k0 >> 15                   (a per-node key, k1 odd; ten 32-bit operations)
 *   bvg_arc_mix(x, y) = k1 * y + k0                                      (mod 2^64)
 * chk = sum of bvg_arc_mix over all arcs, mod 2^64: commutative, so node-range shards reduce with
 * a plain sum (one RCCL all-reduce of {arcs, chk}).  Linear in y under the node's key: a kernel pays
 * one 32 x 32 + 64 multiply-add per successor (the reference's SpeedTest.java:127-135 does nothing
 * with them) and adds d * k0 per node.
 * WHAT THE SCAN MUST DO (round 6, the contract the timed region is held to): ONE multiply-add per
 * PRODUCED successor -- every decoded residual, every element of an interval, every element a copy
 * block keeps, leaf or stored list alike, is read (or generated) and multiplied on its own.  The
 * linear form has closed forms over runs (len * left + len (len - 1) / 2 for an interval, a
 * difference of prefix sums for a kept block): the kernels never use one -- that would be work
 * skipped.  The only per-node constants folded are d * k0 and d * k1 * base (base = the node base /
 * block base every id of the block is stored relative to).  tests/test_gpu_checksum_integrity.py
 * changes ONE element of a stored list and checks that every node copying it moves by exactly
 * k1(node) * delta, through the lean kernel.
 * WHAT IT DETECTS: a single wrong, missing, surplus or misattributed successor changes the sum
 * unless k1 * dy + (0 or k0) == 0 mod 2^64 -- never for one wrong value with |dy| < 2^63 (k1 is
 * odd), with probability ~2^-32 for a missing / surplus / misattributed one (32-bit keys; nodes
 * whose 32-bit hashes collide share a key).  Two errors inside ONE node that cancel in the sum of
 * its successors are invisible: the materialising parity tests (every successor against the oracle
 * / the golden) are the primary gate, this is the scan's self-check; bench.py's untimed gate also
 * materialises one tile through bvg_decode_range: every successor against the oracle's.  (Rounds 1-4 used a non-linear
 * mix: six vector instructions per arc, a quarter of what a decoded successor has to cost.) */
uint64_t bvg_arc_mix(uint64_t x, uint64_t y);

#ifdef __cplusplus
}
#endif
#endif /* BVGRAPH_HIP_H */
