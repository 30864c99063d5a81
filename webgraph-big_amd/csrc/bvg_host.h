// bvg_host.h — what the host-side translation units of libbvgraph_hip.so share (round 6: csrc/bvg_api.hip split into plan / index / tier scheduler / C entry points):
// the handle, the block plan, the residual skip index and the functions that cross the files.  Internal: nothing here is part of the C ABI (include/bvgraph_hip.h).
//   bvg_plan.hip    parameters, handles, the block plan (node blocks of ~4 KiB of stream, halos), the packed offsets, opening a graph
//   bvg_index.hip   (kernels) + bvg_index_host.hip: the residual skip index -- granularity, the build (counting pass, dense walk, validating pass), basename.bvgidx on disk
//   bvg_sched.hip   run_decode: the tier scheduler (tier 0 + LDS classes + giants launched side by side, fail-over, what a scan learns about its blocks)
//   bvg_api.hip     the extern "C" entry points
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include "bvg_kernels.h"

using namespace bvg;

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            if (dbg_on()) fprintf(stderr, "[bvg] %s -> %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? BVG_E_NOMEM : BVG_E_HIP;                         \
        }                                                                                     \
    } while (0)


namespace bvghost {


constexpr uint32_t kDefaultBlockBits = 32768;   // ~4 KiB of compressed stream per wavefront
constexpr uint64_t kPad = 64;                   // zero bytes after the stream (8-byte loads + record overruns)
constexpr uint32_t kGiantResident = 512;        // giant workgroups (512 threads, 88 registers: 2 wavefronts per SIMD each) that can be resident at once: 2 per CU
constexpr uint32_t kGiantSlots = 768;           // their work areas: half as many again (a free one always turns up)
static uint32_t giant_slots() { if (knob("BVG_GSLOTS")) { const int v = atoi(knob("BVG_GSLOTS")); if (v >= 1 && v <= 8192) return (uint32_t)v; } return kGiantSlots; }   // (experiments)

// a device allocation freed on every return path
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    int alloc(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 1) == hipSuccess) return 0; p = nullptr; (void)hipGetLastError(); return 1; }
    void* release() { void* q = p; p = nullptr; return q; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// Residual skip index of the plan blocks [blk_lo, blk_hi) (a shard builds only its own blocks; everything outside has no entries and
// is decoded index-less).  Also the record of which blocks a VALIDATING pass of the row kernel has decoded from end to end
// (fmt[b] == 1): the lean scan kernel (bvg_scan.hip) takes only those.  Immutable once published.
struct SkipIndex {
    int device = 0;
    uint32_t blk_lo = 0, blk_hi = 0;
    uint64_t total = 0; uint64_t* d_first = nullptr; uint16_t* d_bit = nullptr; void* d_val = nullptr; uint8_t* d_fmt = nullptr;
    bool wide = false;                        // entries hold 64-bit values (built by the 64-bit kernels); a handle running the other width ignores the index
    uint32_t skip_min = kSkipMin, skip_shift = 4;   // granularity: lists of >= skip_min residuals hold one entry per 2^skip_shift residuals (skip_granularity() when it is built)
    bool failed = false;                      // the build of [blk_lo, blk_hi) failed: no arrays; scans of those blocks run index-less.  WHY it failed decides what happens next:
    enum { kStream = 1, kResources = 2 };     //   a stream the checking kernels refuse stays refused (only bvg_build_index tries again); running out of memory (or any other HIP
    int fail_cause = 0;                       //   error) is transient: the scans try again every kRetryEvery-th time.  Several failed ranges (two shards that alternate) are kept
    struct FailedRange { uint32_t lo, hi; int cause; };         // side by side, EACH WITH ITS OWN CAUSE (round 6), so that neither pays its counting pass again because of the
    std::vector<FailedRange> failed_ranges;                     // other, and a range that ran out of memory is retried whatever made another one fail (fail_cause: the latest)
    mutable std::atomic<uint32_t> backoff{0}; // scans left before the next automatic attempt (a failed snapshot with kResources; a good partial one whose whole-graph rebuild failed)
    static constexpr uint32_t kRetryEvery = 8;
    bool covers(uint32_t lo, uint32_t hi) const {
        if (!failed) return blk_lo <= lo && hi <= blk_hi;
        for (const auto& r : failed_ranges) if (r.lo <= lo && hi <= r.hi) return true;
        return false;
    }
    int cause_of(uint32_t lo, uint32_t hi) const {            // why the failed range that covers [lo, hi) failed (0: none does)
        for (const auto& r : failed_ranges) if (r.lo <= lo && hi <= r.hi) return r.cause;
        return 0;
    }
    uint64_t gen = 0;                         // identity of this snapshot: what a handle learned about blocks (tier lists, lean / row split) holds for ONE snapshot only
    std::vector<uint64_t> h_first;            // nblk + 1 entry indices (host copy: index_bytes of a range)
    std::vector<uint8_t> h_fmt;               // host copy of d_fmt: 1 = validated by the row kernel (the lean scan kernel may take the block)
    SkipIndex() = default; SkipIndex(const SkipIndex&) = delete; SkipIndex& operator=(const SkipIndex&) = delete;
    ~SkipIndex() {
        (void)hipSetDevice(device);
        if (d_first) (void)hipFree(d_first);
        if (d_bit) (void)hipFree(d_bit);
        if (d_val) (void)hipFree(d_val);
        if (d_fmt) (void)hipFree(d_fmt);
    }
};

struct Plan {
    uint32_t block_bits = 0;
    uint32_t nblk = 0;
    uint64_t* d_first = nullptr; uint32_t* d_halo = nullptr; uint64_t* d_mask = nullptr;
    std::vector<uint64_t> h_first;
    std::vector<uint32_t> h_maxd;             // largest (list + the W lists before it) a block decodes: predicts its tier
    uint64_t version = 0;
    // residual skip index: an immutable snapshot (SkipIndex below), replaced as a whole and read through atomic_load, so a scan
    // running on another thread keeps the arrays it started with
    std::shared_ptr<struct SkipIndex> skip;
    void release() {
        std::atomic_store(&skip, std::shared_ptr<struct SkipIndex>());
        if (d_first) (void)hipFree(d_first);
        if (d_halo) (void)hipFree(d_halo);
        if (d_mask) (void)hipFree(d_mask);
        d_first = nullptr; d_halo = nullptr; d_mask = nullptr; nblk = 0; h_first.clear(); h_maxd.clear();
    }
    int device = 0;
    Plan() = default;
    Plan(const Plan&) = delete;
    Plan& operator=(const Plan&) = delete;
    ~Plan() { (void)hipSetDevice(device); release(); }
};

struct Shared {
    int device = 0;
    bvg_params p{};
    uint8_t* d_graph = nullptr; uint64_t nbytes = 0; uint64_t padded = 0; bool own_graph = false;
    // the offsets index: packed (owned: 4 bytes per node + 8 per 2^kOffShift nodes) or, as a fallback, the plain 64-bit array
    Offsets offs{nullptr, nullptr, nullptr};
    uint32_t* d_off_lo = nullptr; uint64_t* d_off_hi = nullptr; uint64_t* d_off_wide = nullptr; bool own_wide = false;
    uint64_t offsets_bytes() const { return offs.lo ? ((uint64_t)p.nodes + 1) * 4 + ((((uint64_t)p.nodes + 1) >> kOffShift) + 1) * 8 : ((uint64_t)p.nodes + 1) * 8; }
    uint64_t total_bits = 0;
    bool wide = false;
    // Block plans are immutable once built and shared by reference count: a handle holds the one it decodes with for the whole
    // call, so a bvg_copy() flyweight asking for another block size (bvg_set_tuning) on another thread can never free arrays
    // under a kernel in flight.  At most one plan per block size is kept; a new size evicts the others from the table (they
    // live on until their last user returns).  The residual skip index belongs to its plan and is published through
    // Plan::skip_state (release / acquire).
    std::map<uint32_t, std::shared_ptr<Plan>> plans; std::mutex mu; std::mutex skip_mu;
    // cached shard bounds (bvg_shard_bounds): key = (k << 2) | balance
    std::map<uint64_t, std::vector<int64_t>> shard_bounds; std::mutex shard_mu;
    std::atomic<int> refs{1};
};


}  // namespace bvghost
using namespace bvghost;

struct bvg_graph {
    Shared* sh = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* d_acc = nullptr;      // 4 result words + 8 debug counters
    uint32_t* d_fail = nullptr;               // [0] count, [1..] list
    uint32_t fail_cap = 0;
    uint64_t node_base = 0;
    bvg_tuning tun{};
    void* slow_ws = nullptr; uint64_t slow_ws_bytes = 0;   // tier-2 (global-memory) pools, kept between calls
    // predicted tiers run concurrently with tier 0 on high-priority side streams (their few, long blocks are the critical path)
    static constexpr int kSide = 5;            // [0] giants (global-memory kernel), [1..4] one per LDS size class
    hipStream_t side[kSide] = {}; hipEvent_t side_ev[kSide] = {};
    void* giant_ws = nullptr; uint64_t giant_ws_bytes = 0; uint32_t* d_gslots = nullptr;   // work areas of the giant kernel: kGiantSlots slots + their busy flags
    void* flow_ws = nullptr; size_t flow_ws_bytes = 0; uint32_t flow_waves = 0;   // scratch of the flow scan kernel (bvg_flow.hip): one slice per resident wavefront
    void* dr_ws = nullptr; size_t dr_ws_bytes = 0;   // bvg_decode_range / bvg_successors_batch workspace, kept between calls (grown on demand)
    void* tr_ws = nullptr; size_t tr_ws_bytes = 0;   // bvg_transpose workspace, kept between calls
    size_t tr_o_cum = 0, tr_o_succ = 0;             // where the last transpose left the graph's own CSR in it (bvg_symmetrize)
    int skip_mode = 0; uint32_t* skip_cnt = nullptr;   // transient: set while this handle builds the skip index
    std::shared_ptr<SkipIndex> skip_building;          // transient: the index the fill pass (skip_mode 2) writes
    struct Pred {
        uint64_t plan_version = 0, skip_gen = 0; uint32_t lo = 0, n = 0, pool0 = 0, mode = 0; uint32_t* d_lists = nullptr; uint32_t count[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t giant_need = 0;
        std::vector<uint8_t> learned; std::vector<uint8_t> leanfail; uint64_t learned_version = 0, learned_gen = 0; uint32_t learned_pool0 = 0, learned_mode = 0; bool dirty = false;   // tier in which a mispredicted block finally succeeded: the next scans send it there directly
    } pred2[2];                                      // [0] scans, [1] materialising calls (round 6: a handle that alternates bvg_scan and bvg_decode_range keeps what it learned for each; one slot made every change of mode start from the prediction again)
};

namespace bvghost {

// No C++ exception may cross the C ABI (a JVM behind JNI would be torn down by std::terminate): entry points that allocate
// host memory run inside this guard.
template <typename F> static int guarded(F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return BVG_E_NOMEM; }
    catch (const std::length_error&) { return BVG_E_ARG; }
    catch (...) { return BVG_E_STATE; }
}

// the host's own bit reader (properties-side decoding of .offsets: bvg_decode_offsets, BVGraph.java:870-898)
struct HostBits {
    const uint8_t* p; uint64_t nbits, pos = 0; bool eof = false;
    uint64_t peek() const {
        uint64_t byte = pos >> 3, nb = nbits >> 3; uint64_t hi = 0; uint8_t nx = 0;
        for (int i = 0; i < 8; i++) hi = (hi << 8) | (byte + i < nb ? p[byte + i] : 0);
        nx = byte + 8 < nb ? p[byte + 8] : 0;
        unsigned sh = (unsigned)(pos & 7);
        return sh ? (hi << sh) | ((uint64_t)nx >> (8 - sh)) : hi;
    }
    uint64_t bits(unsigned n) { if (!n) return 0; uint64_t w = peek(); pos += n; if (pos > nbits) eof = true; return w >> (64 - n); }
    uint64_t unary() {
        uint64_t z = 0;
        for (;;) {
            uint64_t w = peek();
            if (w) { unsigned lz = (unsigned)__builtin_clzll(w); pos += lz + 1; if (pos > nbits) eof = true; return z + lz; }
            pos += 64; z += 64;
            if (pos >= nbits) { eof = true; return z; }
        }
    }
    uint64_t gamma() { uint64_t m = unary(); if (m > 63) { eof = true; return 0; } return ((1ull << m) | bits((unsigned)m)) - 1; }
    uint64_t delta() { uint64_t m = gamma(); if (m > 63) { eof = true; return 0; } return ((1ull << m) | bits((unsigned)m)) - 1; }
};

struct PackedOffsets { uint32_t* lo; uint64_t* hi; };   // bvg_tile hands over an index it wrote in packed form

// ---- bvg_plan.hip
uint64_t next_plan_version();
uint32_t block_bits_of(const bvg_graph* g);
int open_common(const bvg_params* p, const uint8_t* h_graph, const void* d_graph_in, uint64_t nbytes, const uint64_t* h_offsets,
                const void* d_offsets_in, int device, bvg_graph** out, const PackedOffsets* packed = nullptr);
Codings codings_of(const bvg_params& p);
int check_params(const bvg_params& p);
int read_file(const std::string& path, std::vector<uint8_t>& out);
int make_handle(Shared* sh, bvg_graph** out);
void release_shared(Shared* sh);
int ensure_device(int device);
int build_plan(bvg_graph* g, uint32_t block_bits, std::shared_ptr<Plan>& out);
int read_offset(const Shared* sh, int64_t x, uint64_t* out);
int pack_offsets(Shared* sh, const uint64_t* src_dev, const uint64_t* src_host);

// ---- bvg_sched.hip
// Runs the decode kernel over the blocks intersecting [from,to); slow-path relaunches included.
// `batch` != nullptr: the blocks are the even entries of a per-call plan (one request each, bvg_successors_batch).
struct BatchPlan { const uint64_t* d_first; const uint32_t* d_halo; const uint64_t* d_mask; uint32_t requests; };

int run_decode(bvg_graph* g, int64_t from, int64_t to, bool materialise, const uint64_t* d_cum, int64_t* d_succ, int32_t* d_outdeg,
               bvg_scan_result* res, const BatchPlan* batch = nullptr, const std::shared_ptr<Plan>* use_plan = nullptr);

// The granularity of a graph's skip index: lists of >= `smin` residuals hold one entry per 2^shift residuals.  A residual pass lasts as long as its longest task, so the
// threshold matters as much as the spacing: 16 / 16 (a list of 16-23 residuals is two tasks instead of one of up to 23 steps) gains on every shape over rounds 1-3's 24 / 16
// -- w0 +7.2 %, uk +3.8 %, web +2.6 %, eu +1.5 %, eu15 +1.0 % -- for 0.1-8 % more entries.  A sparse graph's pass holds few tasks, and one entry per 8 residuals from
// lists of 8 on shortens it further: web +11.5 %, uk +5.1 %, cnr-2000 +2.3 % over 24 / 16, for 0.1-0.3 GB of entries per GB of stream; on the dense default workload 8 / 8
// is no faster than 16 / 16 and takes +80 % of an index that is half the stream already, on the reference-free w0 neither (its lists are residuals only: +30 % of resident
// bytes) -- profiles/r04_skipgran3.txt.  So: 8 / 8 below 40 arcs per node (128 bits per node when the arc count is unknown) when the graph has references, else 16 / 16.

// ---- bvg_index_host.hip
void skip_granularity(const Shared* sh, uint32_t& smin, uint32_t& shift);
int build_skip(bvg_graph* g, const std::shared_ptr<Plan>& plp, uint32_t blo, uint32_t bhi, bool retry_failed = false, bvg_scan_result* first_scan = nullptr, int64_t sfrom = 0, int64_t sto = 0, bool* first_scan_done = nullptr);
int save_index_impl(bvg_graph* g, const char* path);
int load_index_impl(bvg_graph* g, const char* path);

}  // namespace bvghost
