// bvg_kernels.h — launch interface between the host API (bvg_api.hip) and the kernels (bvg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "bvg_device.h"

namespace bvg {

// Environment switches force tiers / emission forms for the parity tests and the experiments under profiles/.  A product process
// never looks at them: they are live only when BVG_TEST_KNOBS is set when the library is first used (tests/conftest.py and the
// profiling scripts set it).  BVG_DEBUG (per-tier timing lines on stderr) is honoured always, read once.
inline const char* knob(const char* name) { static const bool live = getenv("BVG_TEST_KNOBS") != nullptr; return live ? getenv(name) : nullptr; }
inline bool dbg_on() { static const bool on = getenv("BVG_DEBUG") != nullptr; return on || knob("BVG_DEBUG") != nullptr; }

// LDS geometry of the fast (one wavefront per node block) decode kernel.
#ifndef BVG_SKIP_MIN
#define BVG_SKIP_MIN 16
#endif
#ifndef BVG_SKIP_EVERY
#define BVG_SKIP_EVERY 16
#endif
#ifndef BVG_RES_UNROLL
#define BVG_RES_UNROLL 2
#endif
constexpr uint32_t kResUnroll = BVG_RES_UNROLL;   // residual segments decoded per lane and pass, interleaved (row kernel)
#ifndef BVG_SHORT_TASK
#define BVG_SHORT_TASK 6
#endif
constexpr uint32_t kShortTask = BVG_SHORT_TASK;   // residual tails this short are dealt after the long tasks of the row
constexpr uint32_t kSkipMin = BVG_SKIP_MIN, kSkipEvery = BVG_SKIP_EVERY;   // residual skip index granularity (kSkipEvery: a power of two)
constexpr uint32_t kAccStripes = 2048, kAccStride = 32;   // result stripes (power of two), 256 bytes apart (stripe 0 also carries 16 debug counters)
constexpr int kRing = 128;           // node-metadata ring (node id mod kRing); supports window sizes <= kMaxWindow
constexpr int kMaxWindow = 64;       // window sizes up to here run in the LDS row kernels
constexpr int kMaxHalo = 64;         // halo nodes a block may need from before its first node (one row), selected by a 64-bit mask
// Larger windows (BVGraph allows any; LAW's "highly compressed" stores use ~70): every block takes the generic global-memory
// kernel, whose node ring is kRingBig entries, and a block's halo is then a plain count (every halo node is decoded).
constexpr int kRingBig = 2048;
constexpr int kMaxWindowBig = kRingBig - 64;
constexpr int kMaxHaloBig = 8192;

// Offsets index in HBM, packed: one 64-bit base per 2^kOffShift nodes + a 32-bit distance per node (4 bytes per node instead of
// the 8 of a plain array; the reference keeps it as an Elias-Fano list, BVG:1545-1558).  `wide` is the plain form: the fallback
// when 2^kOffShift consecutive records span 2^32 bits or more, and BVG_WIDE_OFFSETS=1.
constexpr int kOffShift = 10;
struct Offsets {
    const uint32_t* lo; const uint64_t* hi; const uint64_t* wide;
    __host__ __device__ __forceinline__ uint64_t operator[](int64_t x) const { return lo ? hi[x >> kOffShift] + lo[x] : wide[x]; }
};
// packs entries [first, first + count) of an index (first: a multiple of 2^kOffShift; src[0] is entry `first`); *overflow != 0 if a distance does not fit
void launch_pack_offsets(const uint64_t* src, int64_t first, int64_t count, uint32_t* lo, uint64_t* hi, unsigned* overflow, hipStream_t s);
void launch_narrow_succ(const int64_t* in, uint32_t* out, uint64_t n, hipStream_t s);   // int64 successors -> uint32 (host path of graphs with <= 2^32 nodes)
void launch_unpack_offsets(Offsets o, int64_t first, int64_t count, uint64_t* dst, hipStream_t s);

struct DecodeArgs {
    const uint8_t* graph; uint64_t limit_byte;
    uint64_t padded_bytes;              // readable bytes of `graph` (multiple of 16)
    Offsets offsets;                    // n+1 bit positions
    int64_t n;
    int64_t from, to;                   // only nodes in [from,to) are reported
    const uint64_t* blk_first;          // nblk+1 node ids
    const uint32_t* blk_halo;           // halo length (<= kMaxHalo)
    const uint64_t* blk_mask;           // bit j = node (first-1-j) is needed
    const uint32_t* work_list;          // block ids to run, or nullptr = blockIdx.x + blk_lo
    uint32_t blk_lo;
    int window, min_interval;
    Codings cod;
    uint64_t node_base;
    uint64_t wide_half;                 // lean scan kernel on graphs beyond 2^32 nodes: the block base lies this far below the block's first node (2^31; tests shrink it)
    unsigned long long* acc;            // [0] arcs [1] chk [2] nodes [3] error bits (+ debug counters) of stripe 0
    uint32_t acc_mask;                  // the block results are striped over acc_mask+1 copies of those four words, kAccStride words apart
                                        // (one address for every block costs ~12 ns per atomic: 9 ms per GiB of 4 KiB blocks), summed by reduce_acc
    // materialise
    const uint64_t* cum;                // exclusive prefix of outdegrees for nodes [from,to], or nullptr
    int64_t* succ; int32_t* outdeg;
    // slow path hand-off
    uint32_t* fail_list; uint32_t* fail_count; uint32_t fail_cap;
    uint32_t* fail_need;                // per failed block: list-pool elements it would need (0xFFFFFFFF = unknown / other cause)
    // slow-path pools (global memory), per workgroup
    void* gpool; uint64_t gpool_elems; void* gscr; uint64_t gscr_elems;
    // giant kernel: the work areas are SLOTS shared by the whole launch (as many as workgroups can be resident, not one per block): a workgroup takes a free
    // one when it starts (atomicCAS on gslots[i]) and hands it back when it ends; nullptr = area blockIdx.x (batched launches)
    uint32_t* gslots; uint32_t gnslots;
    // fast path: LDS pool / scratch sizes in elements (dynamic shared memory)
    uint32_t lds_pool_elems, lds_scr_elems;
    uint32_t lds_stage_words;           // row-static kernel: LDS window over the stream, in dwords (multiple of 4)
    uint32_t grab_threshold;            // stream kernel: idle lanes that trigger a batched grab
    uint32_t batch;                     // 1 = bvg_successors_batch: block 2i is request i; outputs are indexed by request
    uint32_t pass_cost;                 // task emission: assumed fixed cost of one level pass, in merge steps (per-row choice of the emission form)
    uint32_t emit_tasks;                // 1 = row kernel variant with level-synchronous task emission (chosen per row), 0 = pipelined loop only
    uint32_t dbg;                       // timing experiments only (BVG_DBG): 1 skip emission, 2 skip residual decode, 4 skip parse, 16 force task rows, 32 force pipelined rows, 64 work counters, 128 skip the task merge loop
    // residual skip index (row kernel): for every node with >= kSkipMin residuals, one entry per kSkipEvery residuals
    // {bit offset of that residual's code from the record start, value of the residual before it}; entries of a block
    // are contiguous, in node order.  skip_mode 1 = count entries per block, 2 = fill them (and validate the block), 3 = validate the block decoding
    // WITH the entries a dense walk has just filled, checking every one (the row kernels; the giant kernel fills its own in this pass), 0 = use them when present.
    const uint64_t* skip_first;         // nblk+1 entry indices, or nullptr
    uint16_t* skip_bit; void* skip_val; // entries: 16-bit bit offset (a record that uses the index fits the LDS window, <= 64 Kbit) + one successor-typed value (4 or 8 bytes)
    uint32_t* skip_cnt;                 // skip_mode 1: per-block entry count out
    uint32_t skip_min, skip_shift;      // the granularity of THIS index: lists of >= skip_min residuals hold one entry per 2^skip_shift residuals (chosen per graph when the index
                                        // is built, bvg_api.hip skip_granularity; kSkipMin / kSkipEvery are the dense graphs' values and what a scan without index is handed)
    uint8_t* skip_fmt;                  // per block: who filled its entries (skip_mode 2) -- 1 = row kernels (one slot per entry), 2 = the giant
                                        // kernel (two slots per entry: 32-bit offsets); a kernel uses only entries of its own format
    uint32_t skip_mode;
    uint32_t xcds;                      // XCDs the work order is laid out for (8 on MI355X; 1 = plain order): xcd_order(), bvg_rows_common.h
    uint32_t flat_recs;                 // flat scan kernel (experimental/bvg_flat.hip): records per super-row (a multiple of 64: 64 on dense graphs, up to 256 on sparse ones)
};

void launch_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, bool slow, hipStream_t s);
// the lean LDS-resident row kernel (bvg_rows.hip): tiers 0 and 1
void launch_rows_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, hipStream_t s);
// the lean scan kernel (bvg_scan.hip): validated blocks of a scan or (materialise) of a decode into a.succ, 32-bit lists, default codings, skip index present
void launch_scan_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, int occ, bool materialise, hipStream_t s);   // occ: 4 / 5 / 6 wavefronts per SIMD (128 / 96 / 80 VGPRs)
size_t scan_static_lds();

// the offsets index from a bare .graph in parallel (bvg_derive.hip): chunks of the stream walked speculatively, one code per lane and step, and iterated
// to the one consistent walk; 0 = done (err[0] != 0 on a bad stream), < 0 = not applicable / gave up: fall back to launch_derive_offsets
int derive_offsets_parallel(const uint8_t* graph, uint64_t nbytes, int64_t n, int window, int min_interval, Codings cod, uint64_t* offsets, unsigned* err,
                            hipStream_t s, int* rounds);

// Experiments that lost to the row kernel are compiled only by `make experimental` (-DBVG_EXPERIMENTAL): the row kernel
// with one workgroup of nw wavefronts per block sharing the pool (bvg_rows_wg.hip), the streaming data-flow kernel
// (bvg_stream.hip), the flow scan kernel (bvg_flow.hip), 
#ifdef BVG_EXPERIMENTAL
constexpr bool kExperimental = true;
void launch_rows_wg_decode(const DecodeArgs& a, uint32_t nblocks, int nw, hipStream_t s);
size_t rows_wg_static_lds(int nw);
void launch_stream_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, hipStream_t s);
size_t flow_scratch_bytes_per_wave(int window);
size_t flow_lds_bytes(uint32_t ring_cap);
void launch_flow_scan(const DecodeArgs& a, uint32_t nblocks, uint32_t waves, void* scratch, uint32_t ring_cap, hipStream_t s);
// the flat scan kernel (experimental/bvg_flat.hip, round 5): scan_kernel's blocks with per-record state in an LDS table and every pass a flat task list over
// 64-256 records; bit-exact, slower: BVG_FLAT=1 selects it in the experimental build
void launch_flat_decode(const DecodeArgs& a, uint32_t nblocks, bool many_waves, hipStream_t s);
size_t flat_table_bytes(uint32_t recs, int window);
#else
constexpr bool kExperimental = false;
inline void launch_rows_wg_decode(const DecodeArgs&, uint32_t, int, hipStream_t) {}
inline size_t rows_wg_static_lds(int) { return 0; }
inline size_t flow_scratch_bytes_per_wave(int) { return 0; }
inline size_t flow_lds_bytes(uint32_t) { return 0; }
inline void launch_flow_scan(const DecodeArgs&, uint32_t, uint32_t, void*, uint32_t, hipStream_t) {}
inline void launch_flat_decode(const DecodeArgs&, uint32_t, bool, hipStream_t) {}
inline size_t flat_table_bytes(uint32_t, int) { return 0; }
#endif

// tier 2a (bvg_giant.hip): blocks with lists / records too large for LDS, one 256-thread workgroup per block, work areas as for the
// generic kernel (a.gpool / a.gscr); default codings and windows <= kMaxWindow only
void launch_giant_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, hipStream_t s);

// sums the result stripes into stripe 0 (one workgroup)
// the filling pass of the skip index as a dense walk (bvg_index.hip): entries of the blocks of the work list, where the counting pass allotted them
void launch_index_walk(const DecodeArgs& a, uint32_t nblocks, bool wide, hipStream_t s);
void launch_clear_unmarked_entries(const uint64_t* first, const uint8_t* fmt, uint16_t* bit, void* val, uint32_t blo, uint32_t bhi, bool wide, hipStream_t s);
void launch_reduce_acc(unsigned long long* acc, uint32_t stripes, hipStream_t s);
// *out += position-keyed 64-bit hash of the nbytes at p (device memory, 8-byte aligned); *out must be zeroed by the caller
void launch_hash_words(const void* p, uint64_t nbytes, unsigned long long* out, hipStream_t s);
// thread per node: outdegree (BVG:821-842)
void launch_outdegrees(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, int64_t from, int64_t to,
                       int outdegree_coding, int32_t* out, unsigned long long* total, hipStream_t s);
// outdegree of an arbitrary list of nodes (random access, BVG:821-842); also writes the two plan entries {x, x+1} per request
void launch_outdegrees_gather(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, const int64_t* nodes, int64_t count,
                              int outdegree_coding, int32_t* out, uint64_t* first, hipStream_t s);
// exclusive prefix sum of int32 -> uint64 (n+1 outputs), single stream, hand-written 3-phase scan
void launch_exclusive_scan(const int32_t* in, uint64_t* out, int64_t n, uint64_t* tmp, hipStream_t s);
size_t scan_tmp_elems(int64_t n);

// plan: block boundaries at ~equal compressed bits, then per-boundary halo (reference-chain walk)
void launch_plan_boundaries(Offsets offsets, int64_t n, uint64_t block_bits, uint64_t nb, uint64_t* first, hipStream_t s);
// (window > kMaxWindow: halo = distance to the farthest node reached, up to kMaxHaloBig, mask = all ones)
void launch_plan_halo(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, int64_t n, const uint64_t* first, uint32_t nblk,
                      int window, Codings cod, uint32_t* halo, uint64_t* mask, hipStream_t s);

// plan: largest outdegree among the nodes a block decodes (its own + its halo): predicts the LDS tier it needs
void launch_plan_longest(Offsets offsets, const uint64_t* first, uint32_t nblk, uint64_t* node, uint64_t* bits, hipStream_t s);
void launch_plan_maxd(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, const uint64_t* first, const uint32_t* halo, uint32_t nblk,
                      int outdegree_coding, int window, uint32_t* maxd, uint64_t* bign, uint32_t* bigd, hipStream_t s);

// offsets index from a bare .graph (BVGraph -O / writeOffsets, BVG:2595-2609; loadSequential/loadOffline, BVG:1345-1464):
// one wavefront parses the stream sequentially (all lanes in step, LDS-staged); offsets[n+1] out, err[0] != 0 on a bad stream
void launch_derive_offsets(const uint8_t* graph, uint64_t padded_bytes, uint64_t nbytes, int64_t n, int window, int min_interval, Codings cod,
                           uint64_t* offsets, unsigned* err, hipStream_t s);

// BVGraph.store on the device (bvg_encode.hip): adjacency in CSR form (device pointers) -> .graph bytes + offsets (hipMalloc'ed here)
int encode_store_dev(const bvg_params& p, const uint64_t* d_adj_off, const int64_t* d_adj, int64_t n, int64_t chunk_nodes, hipStream_t s,
                     uint8_t** d_graph_out, uint64_t* graph_bytes, uint64_t** d_offsets_out);

// synthetic workloads (bvg_tile / bvg_mosaic): the cycle {base 0, ..., base k-1} repeated
constexpr int kMosaicMax = 16;
struct MosaicSrc {
    int k;
    const uint8_t* graph[kMosaicMax]; Offsets offs[kMosaicMax];
    uint64_t bits[kMosaicMax], bit_prefix[kMosaicMax + 1]; int64_t node_prefix[kMosaicMax + 1];
    uint64_t cycle_bits; int64_t cycle_nodes;
};
void launch_mosaic_graph(const MosaicSrc& m, uint8_t* dst, uint64_t dst_bytes, int64_t cycles, hipStream_t s);
void launch_mosaic_offsets(const MosaicSrc& m, int64_t cycles, uint32_t* dst_lo, uint64_t* dst_hi, unsigned* overflow, hipStream_t s);   // packed output

// transposition feed (bvg_transpose.hip): stable radix sort of (target, source) pairs + in-degree prefix
size_t transpose_temp_bytes(uint64_t arcs, int64_t n);
hipError_t transpose_pairs(const uint64_t* cum, int64_t n, uint64_t arcs, const int64_t* succ, int64_t* src, uint64_t* keys_out, void* temp, size_t temp_bytes,
                           uint64_t* toffsets, int64_t* tsucc, unsigned* d_bad, hipStream_t s);

// per-node sorted union of two CSR graphs over the same nodes (Transform.union): count pass, then write pass
void launch_union_count(const uint64_t* acum, const int64_t* asucc, const uint64_t* bcum, const int64_t* bsucc, int64_t n, int32_t* cnt, hipStream_t s);
void launch_union_write(const uint64_t* acum, const int64_t* asucc, const uint64_t* bcum, const int64_t* bsucc, int64_t n, const uint64_t* ocum, int64_t* out, hipStream_t s);

}  // namespace bvg
