// bvg_api.hip — host side of libbvgraph_hip.so: the C ABI of include/bvgraph_hip.h.
//
// Mirrors the load path of the reference (ImmutableGraph.load -> BVGraph.loadInternal,
// BVGraph.java:1479-1574): parse .properties, bring .graph into memory (here: HBM), decode the
// .offsets gaps into an index (here: a device array instead of an Elias-Fano list), then serve
// outdegree / successors / sequential scans — all of which run as HIP kernels (bvg_kernels.hip).
// There is no CPU decode path in this library.
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

namespace {

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
    std::vector<std::pair<uint32_t, uint32_t>> failed_ranges;   // side by side, so that neither pays its counting pass again because of the other.
    mutable std::atomic<uint32_t> backoff{0}; // scans left before the next automatic attempt (a failed snapshot with kResources; a good partial one whose whole-graph rebuild failed)
    static constexpr uint32_t kRetryEvery = 8;
    bool covers(uint32_t lo, uint32_t hi) const {
        if (!failed) return blk_lo <= lo && hi <= blk_hi;
        for (const auto& r : failed_ranges) if (r.first <= lo && hi <= r.second) return true;
        return false;
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

}  // namespace

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
    } pred;
};

namespace {

Codings codings_of(const bvg_params& p) {
    Codings c; c.outdegree = p.outdegree_coding; c.block = p.block_coding; c.residual = p.residual_coding;
    c.reference = p.reference_coding; c.block_count = p.block_count_coding; c.zeta_k = p.zeta_k;
    return c;
}

int check_params(const bvg_params& p) {
    auto in = [](int v, std::initializer_list<int> s) { for (int x : s) if (x == v) return true; return false; };
    if (p.nodes < 0) return BVG_E_ARG;
    if (!in(p.outdegree_coding, {BVG_GAMMA, BVG_DELTA})) return BVG_E_UNSUPPORTED;                       // BVG:655-659
    if (!in(p.reference_coding, {BVG_UNARY, BVG_GAMMA, BVG_DELTA})) return BVG_E_UNSUPPORTED;            // BVG:695-700
    if (!in(p.block_count_coding, {BVG_UNARY, BVG_GAMMA, BVG_DELTA})) return BVG_E_UNSUPPORTED;          // BVG:729-734
    if (!in(p.block_coding, {BVG_UNARY, BVG_GAMMA, BVG_DELTA})) return BVG_E_UNSUPPORTED;                // BVG:759-764
    if (!in(p.residual_coding, {BVG_GAMMA, BVG_ZETA, BVG_DELTA, BVG_GOLOMB, BVG_NIBBLE})) return BVG_E_UNSUPPORTED;  // BVG:788-795
    if (!in(p.offset_coding, {BVG_GAMMA, BVG_DELTA})) return BVG_E_UNSUPPORTED;                          // BVG:628-632
    if (p.window_size < 0 || p.window_size > kMaxWindowBig) return BVG_E_UNSUPPORTED;
    if (p.min_interval_length < 0) return BVG_E_ARG;
    if (p.residual_coding == BVG_ZETA && (p.zeta_k < 1 || p.zeta_k > 32)) return BVG_E_ARG;
    return 0;
}

// Host-side MSB-first reader for the .offsets file only (one-off at load).
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

int read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return BVG_E_IO;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize((size_t)sz);
    if (sz && fread(out.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return BVG_E_IO; }
    fclose(f);
    return 0;
}

int make_handle(Shared* sh, bvg_graph** out) {
    bvg_graph* g = new bvg_graph();
    g->sh = sh;
    HIPCHK(hipSetDevice(sh->device));
    HIPCHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        for (int i = 0; i < bvg_graph::kSide; i++) {
            HIPCHK(hipStreamCreateWithPriority(&g->side[i], hipStreamNonBlocking, knob("BVG_PRIO") ? (atoi(knob("BVG_PRIO")) > 0 ? greatest : atoi(knob("BVG_PRIO")) < 0 ? least : 0) : greatest));
            HIPCHK(hipEventCreateWithFlags(&g->side_ev[i], hipEventDisableTiming));
        }
    }
    HIPCHK(hipEventCreate(&g->ev0));
    HIPCHK(hipEventCreate(&g->ev1));
    HIPCHK(hipMalloc(&g->d_acc, (size_t)kAccStripes * kAccStride * sizeof(unsigned long long)));   // stripe 0 also holds the debug counters [4..19]
    g->fail_cap = 1u << 16;
    HIPCHK(hipMalloc(&g->d_fail, (2 * (size_t)g->fail_cap + 1) * sizeof(uint32_t)));
    *out = g;
    return 0;
}

void release_shared(Shared* sh) {
    if (sh->refs.fetch_sub(1) != 1) return;
    (void)hipSetDevice(sh->device);
    sh->plans.clear();
    if (sh->own_graph && sh->d_graph) (void)hipFree(sh->d_graph);
    if (sh->d_off_lo) (void)hipFree(sh->d_off_lo);
    if (sh->d_off_hi) (void)hipFree(sh->d_off_hi);
    if (sh->own_wide && sh->d_off_wide) (void)hipFree(sh->d_off_wide);
    delete sh;
}

int ensure_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return BVG_E_HIP;
    HIPCHK(hipSetDevice(device));
    return 0;
}

uint64_t next_plan_version() { static std::atomic<uint64_t> v{1}; return v.fetch_add(1); }

// Builds the block plan: boundaries at ~equal compressed bits + per-block halo masks.
int build_plan(bvg_graph* g, uint32_t block_bits, std::shared_ptr<Plan>& out) {
    Shared* sh = g->sh;
    std::lock_guard<std::mutex> lk(sh->mu);
    {
        auto it = sh->plans.find(block_bits);
        if (it != sh->plans.end()) { out = it->second; return 0; }
    }
    struct WallClock { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); ~WallClock() { if (dbg_on()) fprintf(stderr, "[bvg] block plan built in %.3f s (wall clock)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } } wall_clock;
    std::shared_ptr<Plan> np = std::make_shared<Plan>();
    Plan& plan = *np;
    plan.device = sh->device;
    plan.block_bits = block_bits;
    auto publish = [&]() { sh->plans.clear(); sh->plans[block_bits] = np; out = np; return 0; };
    const int64_t n = sh->p.nodes;
    if (n == 0) { plan.nblk = 0; plan.h_first.assign(1, 0); return publish(); }
    const uint64_t limit = sh->nbytes;
    uint64_t nb = (sh->total_bits + block_bits - 1) / block_bits;
    if (nb == 0) nb = 1;
    if (nb > 0x7FFFFFF0ull) return BVG_E_UNSUPPORTED;
    uint64_t* d_first0 = nullptr;
    HIPCHK(hipMalloc(&d_first0, (nb + 1) * sizeof(uint64_t)));
    launch_plan_boundaries(sh->offs, n, block_bits, nb, d_first0, g->stream);
    std::vector<uint64_t> first(nb + 1);
    HIPCHK(hipMemcpyAsync(first.data(), d_first0, (nb + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    (void)hipFree(d_first0);
    // drop empty blocks (a record longer than block_bits spans several targets)
    first[0] = 0;
    std::vector<uint64_t> uniq; uniq.reserve(first.size());
    for (size_t i = 0; i < first.size(); i++) if (uniq.empty() || first[i] != uniq.back()) uniq.push_back(first[i]);
    if (uniq.back() != (uint64_t)n) uniq.push_back((uint64_t)n);
    uint32_t nblk = (uint32_t)(uniq.size() - 1);
    // A record longer than the LDS stream window sends its whole block to the giant kernel, which walks a block node by node with the
    // whole workgroup: the ~50 ordinary nodes that share the block with it cost that kernel more than the long record itself (4.4 G-node
    // run: 157 k such blocks = 2.0 s of a scan whose tier 0 ends after 1.2 s).  Cut the block in front of the long record (it is the
    // block's last node or nearly: the record runs past the block's end), so that the nodes before it stay with the LDS kernels.
    if (!knob("BVG_NO_LONGCUT")) {
        uint64_t *d_f = nullptr, *d_node = nullptr, *d_bits = nullptr;
        HIPCHK(hipMalloc(&d_f, (nblk + 1) * sizeof(uint64_t))); HIPCHK(hipMalloc(&d_node, (size_t)nblk * sizeof(uint64_t))); HIPCHK(hipMalloc(&d_bits, (size_t)nblk * sizeof(uint64_t)));
        HIPCHK(hipMemcpyAsync(d_f, uniq.data(), (nblk + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g->stream));
        launch_plan_longest(sh->offs, d_f, nblk, d_node, d_bits, g->stream);
        std::vector<uint64_t> hn(nblk), hb(nblk);
        HIPCHK(hipMemcpyAsync(hn.data(), d_node, (size_t)nblk * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipMemcpyAsync(hb.data(), d_bits, (size_t)nblk * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        (void)hipFree(d_f); (void)hipFree(d_node); (void)hipFree(d_bits);
        std::vector<uint64_t> cut; cut.reserve(uniq.size() + 1024);
        for (uint32_t k = 0; k < nblk; k++) {
            cut.push_back(uniq[k]);
            if (hb[k] + 128 > 32768 && hn[k] > uniq[k] && hn[k] < uniq[k + 1]) cut.push_back(hn[k]);
        }
        cut.push_back(uniq[nblk]);
        if (cut.size() - 1 <= 0x7FFFFFF0ull) { uniq.swap(cut); nblk = (uint32_t)(uniq.size() - 1); }
    }
    // halo per boundary; boundaries whose reference chains reach further back than kMaxHalo nodes are removed.
    // Two rounds: the first one's per-block list sizes show which blocks owe their LDS class (or the giant kernel) to ONE large list; those
    // are cut in front of that list and 2 W + 1 nodes behind it, so that only the few nodes around it run at the class's low occupancy and
    // the rest of the block goes back to tier 0 (the classes held 12 % of the blocks of the default workload and took 28 % of a scan).
    const bool refine = !knob("BVG_NO_LISTCUT") && sh->p.window_size <= kMaxWindow;
    for (int round = 0; round < 2; round++) {
      bool done = false;
      for (int pass = 0; pass < 2 && !done; pass++) {
        uint64_t* d_first = nullptr; uint32_t* d_halo = nullptr; uint64_t* d_mask = nullptr;
        HIPCHK(hipMalloc(&d_first, (nblk + 1) * sizeof(uint64_t)));
        HIPCHK(hipMalloc(&d_halo, (size_t)nblk * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_mask, (size_t)nblk * sizeof(uint64_t)));
        HIPCHK(hipMemcpyAsync(d_first, uniq.data(), (nblk + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g->stream));
        launch_plan_halo(sh->d_graph, limit, sh->offs, n, d_first, nblk, sh->p.window_size, codings_of(sh->p), d_halo, d_mask, g->stream);
        std::vector<uint32_t> halo(nblk);
        HIPCHK(hipMemcpyAsync(halo.data(), d_halo, (size_t)nblk * sizeof(uint32_t), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        bool any_bad = false;
        for (uint32_t k = 0; k < nblk; k++) if (halo[k] == 0xFFFFFFFFu) { any_bad = true; break; }
        if (!any_bad || pass == 1) {
            if (any_bad) { (void)hipFree(d_first); (void)hipFree(d_halo); (void)hipFree(d_mask); return BVG_E_UNSUPPORTED; }
            // per-block largest "list + window" (one wavefront per block), kept on the host to predict tiers; the block's longest list and its node
            uint32_t* d_maxd = nullptr; uint64_t* d_bign = nullptr; uint32_t* d_bigd = nullptr;
            HIPCHK(hipMalloc(&d_maxd, (size_t)nblk * sizeof(uint32_t)));
            const bool want_cuts = refine && round == 0;
            if (want_cuts) { HIPCHK(hipMalloc(&d_bign, (size_t)nblk * sizeof(uint64_t))); HIPCHK(hipMalloc(&d_bigd, (size_t)nblk * sizeof(uint32_t))); }
            launch_plan_maxd(sh->d_graph, limit, sh->offs, d_first, d_halo, nblk, sh->p.outdegree_coding, sh->p.window_size, d_maxd, d_bign, d_bigd, g->stream);
            std::vector<uint32_t> maxd(nblk), bigd(want_cuts ? nblk : 0); std::vector<uint64_t> bign(want_cuts ? nblk : 0);
            hipError_t e2 = hipMemcpyAsync(maxd.data(), d_maxd, (size_t)nblk * sizeof(uint32_t), hipMemcpyDeviceToHost, g->stream);
            if (e2 == hipSuccess && want_cuts) e2 = hipMemcpyAsync(bign.data(), d_bign, (size_t)nblk * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
            if (e2 == hipSuccess && want_cuts) e2 = hipMemcpyAsync(bigd.data(), d_bigd, (size_t)nblk * sizeof(uint32_t), hipMemcpyDeviceToHost, g->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(g->stream);
            (void)hipFree(d_maxd); if (d_bign) (void)hipFree(d_bign); if (d_bigd) (void)hipFree(d_bigd);
            if (e2 != hipSuccess) { (void)hipFree(d_first); (void)hipFree(d_halo); (void)hipFree(d_mask); return BVG_E_HIP; }
            if (want_cuts) {
                // a block above the tier-0 capacity (about 2 000 elements of "worst list + window" / 2) with one list that is most of it
                const uint64_t W1 = knob("BVG_LISTCUT_BEHIND") ? (uint64_t)atoi(knob("BVG_LISTCUT_BEHIND")) : 2 * (uint64_t)sh->p.window_size + 1;   // (behind the list: W + 1 would do for the nodes that copy from it, but chains through them reach back as well: 8 / 15 / 22 nodes measured 251 / 255 / 254 G edges/s)
                std::vector<uint64_t> cut; cut.reserve(uniq.size() + 1024); size_t ncut = 0;
                for (uint32_t k = 0; k < nblk; k++) {
                    cut.push_back(uniq[k]);
                    const uint64_t md = maxd[k] & 0x7FFFFFFFu;
                    if (md / 2 + 64 > 1800 && bigd[k] >= (knob("BVG_LISTCUT_D") ? (uint32_t)atoi(knob("BVG_LISTCUT_D")) : 500u) && uniq[k + 1] - uniq[k] > 2 * W1 + 8) {
                        if (bign[k] > uniq[k] + 4) { cut.push_back(bign[k]); ncut++; }
                        if (bign[k] + W1 + 4 < uniq[k + 1]) { cut.push_back(bign[k] + W1); ncut++; }
                    }
                }
                cut.push_back(uniq[nblk]);
                if (ncut && cut.size() - 1 <= 0x7FFFFFF0ull) {
                    if (dbg_on()) fprintf(stderr, "[bvg] plan: %zu cuts around large lists (%u blocks before)\n", ncut, nblk);
                    (void)hipFree(d_first); (void)hipFree(d_halo); (void)hipFree(d_mask);
                    uniq.swap(cut); nblk = (uint32_t)(uniq.size() - 1);
                    done = true;                                           // next round on the refined boundaries
                    continue;
                }
            }
            if (dbg_on()) {                                                // how many nodes the blocks decode a second time (their halos)
                std::vector<uint64_t> hm(nblk);
                if (hipMemcpy(hm.data(), d_mask, (size_t)nblk * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
                    uint64_t hn = 0; for (uint32_t k = 0; k < nblk; k++) hn += (uint64_t)__builtin_popcountll(halo[k] ? hm[k] & (halo[k] >= 64 ? ~0ull : ((1ull << halo[k]) - 1ull)) : 0ull);
                    fprintf(stderr, "[bvg] plan: %u blocks, %llu halo nodes (%.1f %% of %lld nodes)\n", nblk, (unsigned long long)hn, 100.0 * (double)hn / (double)n, (long long)n);
                }
            }
            plan.d_first = d_first; plan.d_halo = d_halo; plan.d_mask = d_mask;
            plan.nblk = nblk; plan.h_first = uniq; plan.h_maxd.swap(maxd);
            plan.version = next_plan_version();
            return publish();
        }
        // merge blocks: drop un-cuttable boundaries (the halo of a kept boundary does not depend on the others)
        std::vector<uint64_t> kept; kept.reserve(uniq.size());
        for (uint32_t k = 0; k < nblk; k++) if (halo[k] != 0xFFFFFFFFu || k == 0) kept.push_back(uniq[k]);
        kept.push_back((uint64_t)n);
        uniq.swap(kept); nblk = (uint32_t)(uniq.size() - 1);
        (void)hipFree(d_first); (void)hipFree(d_halo); (void)hipFree(d_mask);
      }
      if (!done) break;
    }
    return BVG_E_UNSUPPORTED;
}

uint32_t block_bits_of(const bvg_graph* g) { return g->tun.block_bits ? g->tun.block_bits : kDefaultBlockBits; }

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
// BVG_SKIP_GRAN="min,every" (test knob) overrides; the kernels take the granularity from the index they are handed (DecodeArgs::skip_min / skip_shift), the file carries it.
static void skip_granularity(const Shared* sh, uint32_t& smin, uint32_t& shift) {
    smin = kSkipMin; shift = 0; while ((1u << shift) < kSkipEvery) shift++;
    const double nodes = (double)std::max<int64_t>(sh->p.nodes, 1);
    const bool sparse = sh->p.arcs > 0 ? (double)sh->p.arcs / nodes < 40.0 : (double)sh->total_bits / nodes < 128.0;
    if (sparse && sh->p.window_size > 0) { smin = 8; shift = 3; }
    if (knob("BVG_SKIP_GRAN")) {
        unsigned m = 0, e = 0;
        if (sscanf(knob("BVG_SKIP_GRAN"), "%u,%u", &m, &e) == 2 && m >= 2 && m <= 4096 && e >= 2 && e <= 64 && (e & (e - 1)) == 0) { smin = m; shift = 0; while ((1u << shift) < e) shift++; }
    }
}

// Residual skip index: nodes with long residual lists get one entry per kSkipEvery residuals, so the row kernel can decode a
// long list as independent segments on otherwise idle lanes.  Two passes of the ordinary decode over the plan blocks [blo, bhi):
// count the entries of every block, prefix-sum on the host, fill.  An index, not a cache: every gap is still decoded from the stream.
// The fill pass is also the VALIDATING pass: a block it decodes from end to end with the position logic (which refuses streams
// that overlap, counts that contradict each other, ...) is marked fmt = 1, and only such blocks are given to the lean scan kernel.
// The result replaces the plan's snapshot; scans that hold the old one keep it alive until they return.
// `first_scan` (with its node range): the scan whose first call builds the index wants {nodes, arcs, chk} of that very range -- the validating pass decodes every
// block of the range anyway, so it reports them, and the caller does not scan a second time (only when the index is built for exactly the scan's blocks).
int build_skip(bvg_graph* g, const std::shared_ptr<Plan>& plp, uint32_t blo, uint32_t bhi, bool retry_failed = false, bvg_scan_result* first_scan = nullptr, int64_t sfrom = 0, int64_t sto = 0, bool* first_scan_done = nullptr) {
    Shared* sh = g->sh;
    std::lock_guard<std::mutex> lk(sh->skip_mu);
    Plan& pl = *plp;
    {
        std::shared_ptr<SkipIndex> cur = std::atomic_load(&pl.skip);
        if (cur && !(cur->failed && retry_failed) && cur->covers(blo, bhi)) return 0;   // another thread built it meanwhile (or failed to: not tried again here)
        if (cur && !cur->failed) { blo = 0; bhi = pl.nblk; first_scan = nullptr; }   // a second range: index the whole graph once and for all (the scan's own range is a part of it: it scans afterwards)
    }
    const uint32_t nblk = pl.nblk;
    if (!nblk || sh->p.nodes == 0 || blo >= bhi) return 0;
    const int64_t nfrom = (int64_t)pl.h_first[blo], nto = (int64_t)pl.h_first[bhi];
    const bool build_wide = sh->wide || g->tun.force_wide;
    std::shared_ptr<SkipIndex> ix = std::make_shared<SkipIndex>();
    ix->device = sh->device; ix->blk_lo = blo; ix->blk_hi = bhi; ix->wide = build_wide; ix->gen = next_plan_version();
    skip_granularity(sh, ix->skip_min, ix->skip_shift);
    auto publish = [&]() { std::atomic_store(&pl.skip, ix); return 0; };
    // no index: the scans run without one.  The failure is PUBLISHED (an empty snapshot of the same block range, unless a good index of
    // other blocks exists already), so that later scans of these blocks do not pay the counting pass again and again; bvg_build_index() retries.
    auto give_up = [&](int cause) {
        (void)hipGetLastError();
        std::shared_ptr<SkipIndex> cur = std::atomic_load(&pl.skip);
        if (!cur || cur->failed) {
            std::shared_ptr<SkipIndex> fx = std::make_shared<SkipIndex>();
            fx->device = sh->device; fx->blk_lo = blo; fx->blk_hi = bhi; fx->wide = build_wide; fx->failed = true; fx->gen = next_plan_version();
            fx->fail_cause = cause; fx->backoff.store(SkipIndex::kRetryEvery);
            if (cur) for (const auto& r : cur->failed_ranges) if (!(blo <= r.first && r.second <= bhi) && fx->failed_ranges.size() < 64) fx->failed_ranges.push_back(r);   // the ranges that failed before stay failed
            fx->failed_ranges.emplace_back(blo, bhi);
            std::atomic_store(&pl.skip, fx);
        } else cur->backoff.store(SkipIndex::kRetryEvery);              // a good index of other blocks exists: its whole-graph rebuild is not tried again on every scan
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true) || dbg_on())                        // once per process, whether or not BVG_DEBUG is set: every scan of these blocks is ~5x slower from here on
            fprintf(stderr, "[bvg] warning: the residual skip index of blocks [%u, %u) could not be built (%s); scans of them run without it%s\n", blo, bhi,
                    cause == SkipIndex::kStream ? "the checking kernels refused the stream" : "out of device memory or a HIP error",
                    cause == SkipIndex::kStream ? " (bvg_build_index() tries again)" : " and try again every 8th time");
        return 0;
    };
    DevBuf cnt_d;
    const auto tb0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count(); };
    if (cnt_d.alloc((size_t)nblk * sizeof(uint32_t)) || hipMemset(cnt_d.p, 0, (size_t)nblk * sizeof(uint32_t)) != hipSuccess) return give_up(SkipIndex::kResources);
    g->skip_mode = 1; g->skip_cnt = (uint32_t*)cnt_d.p; g->skip_building = ix;          // (the counting pass counts in the new index's granularity)
    int r = run_decode(g, nfrom, nto, false, nullptr, nullptr, nullptr, nullptr, nullptr, &plp);
    g->skip_mode = 0; g->skip_cnt = nullptr; g->skip_building.reset();
    const double t_count = since();
    std::vector<uint32_t> cnt(nblk);
    if (!r && hipMemcpy(cnt.data(), cnt_d.p, (size_t)nblk * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) r = BVG_E_HIP;
    if (r) return give_up((r == BVG_E_HIP || r == BVG_E_NOMEM) ? SkipIndex::kResources : SkipIndex::kStream);   // a bad stream surfaces in the caller's own decode
    std::vector<uint64_t> first(nblk + 1, 0);
    for (uint32_t i = 0; i < nblk; i++) first[i + 1] = first[i] + ((i >= blo && i < bhi) ? cnt[i] : 0u);
    const uint64_t total = first[nblk];
    if (hipMalloc(&ix->d_first, (size_t)(nblk + 1) * sizeof(uint64_t)) != hipSuccess || hipMalloc(&ix->d_bit, total * sizeof(uint16_t) + 16) != hipSuccess ||
        hipMalloc(&ix->d_fmt, nblk) != hipSuccess || hipMemset(ix->d_fmt, 0, nblk) != hipSuccess ||
        hipMalloc(&ix->d_val, total * (build_wide ? sizeof(uint64_t) : sizeof(uint32_t)) + 16) != hipSuccess) return give_up(SkipIndex::kResources);
    if (hipMemcpy(ix->d_first, first.data(), (size_t)(nblk + 1) * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return give_up(SkipIndex::kResources);
    // (entries nobody fills -- the allotment of a block that ends in the generic kernel -- read as zero: an index, and its file, are reproducible)
    if (hipMemsetAsync(ix->d_bit, 0, total * sizeof(uint16_t) + 16, g->stream) != hipSuccess || hipMemsetAsync(ix->d_val, 0, total * (build_wide ? sizeof(uint64_t) : sizeof(uint32_t)) + 16, g->stream) != hipSuccess) return give_up(SkipIndex::kResources);
    ix->total = total;
    const double t_alloc = since();
    // Filling: a DENSE WALK writes the entries (bvg_index.hip: one lane per long list, the lists of a block queued together), then the validating pass decodes every block
    // WITH them (skip_mode 3: residual tasks instead of one lane's serial walk per list) and checks each entry against the stream as it goes.  BVG_INDEX_WALK=0: round 3's
    // single pass (the row kernel walks, fills and validates in one go, index-less).
    const bool dense_walk = !(knob("BVG_INDEX_WALK") && atoi(knob("BVG_INDEX_WALK")) == 0);
    double t_walk = t_alloc;
    if (dense_walk) {
        DecodeArgs wa{};
        wa.graph = sh->d_graph; wa.limit_byte = sh->nbytes; wa.padded_bytes = sh->padded; wa.offsets = sh->offs; wa.n = sh->p.nodes; wa.from = nfrom; wa.to = nto;
        wa.blk_first = pl.d_first; wa.blk_halo = pl.d_halo; wa.blk_mask = pl.d_mask; wa.work_list = nullptr; wa.blk_lo = blo;
        wa.window = sh->p.window_size; wa.min_interval = sh->p.min_interval_length; wa.cod = codings_of(sh->p);
        wa.skip_first = ix->d_first; wa.skip_bit = ix->d_bit; wa.skip_val = ix->d_val; wa.skip_min = ix->skip_min; wa.skip_shift = ix->skip_shift;
        launch_index_walk(wa, bhi - blo, build_wide, g->stream);
        if (hipStreamSynchronize(g->stream) != hipSuccess) return give_up(SkipIndex::kResources);
        t_walk = since();
    }
    g->skip_mode = dense_walk ? 3 : 2; g->skip_building = ix;
    const bool report = first_scan != nullptr && !(knob("BVG_FIRST_SCAN_TWICE") && atoi(knob("BVG_FIRST_SCAN_TWICE")));
    r = report ? run_decode(g, sfrom, sto, false, nullptr, nullptr, nullptr, first_scan, nullptr, &plp)       // (the same blocks; only what is REPORTED is clipped to the scan's nodes)
               : run_decode(g, nfrom, nto, false, nullptr, nullptr, nullptr, nullptr, nullptr, &plp);
    g->skip_mode = 0; g->skip_building.reset();
    if (!r && report) {
        // the pass decoded with the entries it was validating: the result says so (run_decode could not know their number yet)
        first_scan->index_entries = total;
        first_scan->index_bytes += total * (2 + (build_wide ? sizeof(uint64_t) : sizeof(uint32_t)));        // (the 9 bytes per block were counted by the pass itself: a.skip_first was set)
        if (first_scan_done) *first_scan_done = true;
    }
    if (r) return give_up((r == BVG_E_HIP || r == BVG_E_NOMEM) ? SkipIndex::kResources : SkipIndex::kStream);
    launch_clear_unmarked_entries(ix->d_first, ix->d_fmt, ix->d_bit, ix->d_val, blo, bhi, build_wide, g->stream);
    if (hipStreamSynchronize(g->stream) != hipSuccess) return give_up(SkipIndex::kResources);
    ix->h_fmt.resize(nblk);
    if (hipMemcpy(ix->h_fmt.data(), ix->d_fmt, nblk, hipMemcpyDeviceToHost) != hipSuccess) return give_up(SkipIndex::kResources);
    ix->h_first.swap(first);
    if (dbg_on()) fprintf(stderr, "[bvg] residual skip index: blocks [%u, %u) of %u, %llu entries, %.1f MiB; wall clock: counting pass %.3f s, prefix + allocation %.3f s, dense walk %.3f s, %s pass %.3f s\n", blo, bhi, nblk, (unsigned long long)total, (double)total * (build_wide ? 10.0 : 6.0) / 1048576.0, t_count, t_alloc - t_count, t_walk - t_alloc, dense_walk ? "validating" : "filling + validating", since() - t_walk);
    return publish();
}

int run_decode(bvg_graph* g, int64_t from, int64_t to, bool materialise, const uint64_t* d_cum, int64_t* d_succ, int32_t* d_outdeg,
               bvg_scan_result* res, const BatchPlan* batch, const std::shared_ptr<Plan>* use_plan) {
    Shared* sh = g->sh;
    int r = 0;
    const bool force_slow = g->tun.force_slow || sh->p.window_size > kMaxWindow;   // wide windows: the generic global-memory kernel only
    static const std::shared_ptr<Plan> no_plan = std::make_shared<Plan>();         // batch calls bring their own per-call plan
    std::shared_ptr<Plan> plp = use_plan ? *use_plan : no_plan;                     // held for the whole call (see Shared::plans)
    if (!batch && !use_plan) { r = build_plan(g, block_bits_of(g), plp); if (r) return r; }
    const bool rows_default = (g->tun.reserved & 0xFF) == 0 && !force_slow;
    const bool force_giant = !force_slow && knob("BVG_GIANT") && atoi(knob("BVG_GIANT")) == 2;   // tests: every block through the giant kernel
    const Plan& pl = *plp;
    const bool wide = sh->wide || g->tun.force_wide;
    // block range
    uint32_t lo = 0, nblocks = 0;
    if (batch) nblocks = batch->requests;
    else {
        const std::vector<uint64_t>& hf = pl.h_first;
        lo = (uint32_t)(std::upper_bound(hf.begin(), hf.end(), (uint64_t)from) - hf.begin());
        lo = lo ? lo - 1 : 0;
        uint32_t hi = (uint32_t)(std::lower_bound(hf.begin(), hf.end(), (uint64_t)to) - hf.begin());
        if (hi > pl.nblk) hi = pl.nblk;
        nblocks = hi > lo ? hi - lo : 0;
    }
    // The residual skip index is built the first time it would pay: a SCAN of >= 4096 nodes indexes the blocks it covers (a shard
    // of a multi-GPU scan builds its own part only; a later scan outside them indexes the whole graph), a materialising call
    // the whole graph once it covers a quarter of it.  bvg_build_index() does the same explicitly.
    if (!batch && rows_default && g->skip_mode == 0 && !knob("BVG_NOSKIP") && !g->tun.no_index && (to - from) >= 4096 && nblocks) {
        std::shared_ptr<SkipIndex> cur = std::atomic_load(&plp->skip);
        bool covered = cur && cur->covers(lo, lo + nblocks), retry = false;
        // a build that failed for want of memory is tried again every kRetryEvery-th scan of its blocks; so is the whole-graph rebuild behind a good partial index
        // (the countdown is shared by every handle of the graph: a compare-exchange, so that two threads at 1 cannot wrap it)
        auto tick = [](const SkipIndex& ix) { uint32_t b = ix.backoff.load(); while (b > 0 && !ix.backoff.compare_exchange_weak(b, b - 1)) {} return b; };   // the value before the tick
        if (cur && covered && cur->failed && cur->fail_cause == SkipIndex::kResources) { if (tick(*cur) <= 1) { covered = false; retry = true; } }
        else if (cur && !covered && !cur->failed && tick(*cur) > 0) covered = true;
        if (!covered && (!materialise || (to - from) >= sh->p.nodes / 4)) {
            bool scanned = false;
            r = materialise ? build_skip(g, plp, 0, pl.nblk, retry) : build_skip(g, plp, lo, lo + nblocks, retry, res, from, to, &scanned);
            if (r) return r;
            if (scanned) return 0;                              // the validating pass of the build WAS this scan (same nodes, the checking kernels: bit-exact by construction)
        }
    }
    std::shared_ptr<SkipIndex> skx0 = g->skip_mode >= 2 ? g->skip_building : (g->skip_mode == 1 ? std::shared_ptr<SkipIndex>() : std::atomic_load(&plp->skip));
    if (skx0 && g->skip_mode == 0 && (skx0->failed || g->tun.no_index)) skx0.reset();                // a failed build left no arrays; bvg_tuning.no_index: this handle scans without it
    const std::shared_ptr<SkipIndex> skx = skx0;                                                       // held for the whole call

    if (nblocks > g->fail_cap) {                            // every block may fail over to the slow path
        (void)hipFree(g->d_fail); g->d_fail = nullptr;
        g->fail_cap = nblocks;
        HIPCHK(hipMalloc(&g->d_fail, (2 * (size_t)g->fail_cap + 1) * sizeof(uint32_t)));
    }
    HIPCHK(hipMemsetAsync(g->d_acc, 0, (size_t)kAccStripes * kAccStride * sizeof(unsigned long long), g->stream));
    HIPCHK(hipMemsetAsync(g->d_fail, 0, sizeof(uint32_t), g->stream));

    DecodeArgs a{};
    a.graph = sh->d_graph; a.limit_byte = sh->nbytes; a.padded_bytes = sh->padded; a.offsets = sh->offs; a.n = sh->p.nodes;
    a.from = from; a.to = to;
    a.blk_first = batch ? batch->d_first : pl.d_first; a.blk_halo = batch ? batch->d_halo : pl.d_halo; a.blk_mask = batch ? batch->d_mask : pl.d_mask;
    a.work_list = nullptr; a.blk_lo = lo; a.batch = batch ? 1u : 0u;
    a.window = sh->p.window_size; a.min_interval = sh->p.min_interval_length; a.cod = codings_of(sh->p);
    a.node_base = g->node_base; a.acc = g->d_acc; a.acc_mask = knob("BVG_NOSTRIPE") ? 0u : kAccStripes - 1; a.cum = d_cum; a.succ = d_succ; a.outdeg = d_outdeg;
    a.fail_list = g->d_fail + 1; a.fail_count = g->d_fail; a.fail_cap = g->fail_cap; a.fail_need = g->d_fail + 1 + g->fail_cap;
    a.dbg = knob("BVG_DBG") ? (uint32_t)strtoul(knob("BVG_DBG"), nullptr, 10) : 0;
#ifndef BVG_PROF
    a.dbg &= (16u | 32u | 64u | 4096u | 8192u | 0xFFFF0000u);             // forcing an emission form (16, 32; 4096 / 8192: scan_kernel's opt-in list builds) and the work counters leave the results alone; the
                                                            // phase-skipping bits (1, 2, 4, 128) exist in the profiling build only
#endif
    // The COUNTING pass of the index build needs the record headers only (a node's entry count follows from its residual count): the
    // row kernels skip the residual decode and the emission there (the same switches the profiling build skips phases with), which
    // turns the first of the two index passes into a header walk.  Pool sizing and every fail-over stay as in the filling pass, so a
    // block is counted in the tier that will fill it.
    if (g->skip_mode == 1) a.dbg |= 3u;
    // Row-kernel variant: splitting lists into tasks pays on dense or reference-free graphs; sparse graphs with reference
    // chains (several short levels per row) are served better by the pipelined node-per-lane loop alone.
    {
        const double avg_d = sh->p.arcs > 0 && sh->p.nodes > 0 ? (double)sh->p.arcs / (double)sh->p.nodes : 16.0;
        a.emit_tasks = (sh->p.window_size == 0 || avg_d >= 8.0) ? 1u : 0u;     // (sparse web shape, 11 arcs a node: the scan kernel still gains 3 %, profiles/r03_web_lean.txt)
        if (knob("BVG_EMIT")) a.emit_tasks = (uint32_t)strtoul(knob("BVG_EMIT"), nullptr, 10) ? 1u : 0u;
        a.pass_cost = knob("BVG_PASSCOST") ? (uint32_t)strtoul(knob("BVG_PASSCOST"), nullptr, 10) : 10u;   // measured: 11-14 merge steps per level pass; the optimum of the estimate is flat over 8-14
    }
    a.skip_mode = (uint32_t)g->skip_mode; a.skip_cnt = g->skip_cnt;
    {   // the granularity of the index in use -- or of the one being built: the counting pass has no arrays yet
        const SkipIndex* gi = g->skip_mode == 1 ? g->skip_building.get() : skx.get();
        a.skip_min = gi ? gi->skip_min : kSkipMin; a.skip_shift = 0; if (gi) a.skip_shift = gi->skip_shift; else while ((1u << a.skip_shift) < kSkipEvery) a.skip_shift++;
    }
    a.xcds = knob("BVG_XCDS") ? (uint32_t)std::max(1, atoi(knob("BVG_XCDS"))) : 8u;
    a.wide_half = knob("BVG_WIDE_HALF") ? strtoull(knob("BVG_WIDE_HALF"), nullptr, 10) : 0x80000000ull;
    if (!batch && rows_default && skx && skx->wide == wide) {
        a.skip_first = skx->d_first; a.skip_bit = skx->d_bit; a.skip_val = skx->d_val; a.skip_fmt = skx->d_fmt;
    }
#ifdef BVG_EXPERIMENTAL
    const bool stream = (g->tun.reserved & 0xFF) == 2;     // A/B switch: the streaming data-flow kernel as tier 0
    const bool legacy = (g->tun.reserved & 0xFF) == 1;     // A/B switch: the generic row kernel (BitCursor) in LDS as tier 0/1
#else
    const bool stream = false, legacy = false;             // (`make experimental` builds the streaming kernel and the generic LDS kernel as tier 0)
#endif
    a.grab_threshold = (g->tun.reserved >> 8) ? (g->tun.reserved >> 8) : 40;
    const size_t esz = wide ? 8 : 4;
    const double avg = sh->p.arcs > 0 && sh->p.nodes > 0 ? (double)sh->p.arcs / (double)sh->p.nodes : 16.0;
    {   // stream window: ~1.5 rows of records, 1..4 KiB (LDS bytes bound occupancy, and occupancy bounds throughput)
        const double bits_per_node = sh->p.nodes > 0 ? (double)sh->total_bits / (double)sh->p.nodes : 64.0;
        uint32_t words = 256;
        while (words < 1024 && (double)words * 32.0 < bits_per_node * 64.0 * 1.5) words *= 2;
        a.lds_stage_words = words;
    }
    if (knob("BVG_STAGE")) a.lds_stage_words = std::min<uint32_t>(std::max<uint32_t>((uint32_t)strtoul(knob("BVG_STAGE"), nullptr, 10) & ~3u, 64u), 2048u);   // (the skip entries hold 16-bit offsets into a record)

    // Workgroup variant of the row kernel (several wavefronts share one pool): scan mode, default codings, 32-bit successors
    int wg_nw = 0;
    {
        const Codings& c = a.cod;
        const bool dflt = c.outdegree == BVG_GAMMA && c.reference == BVG_UNARY && c.block_count == BVG_GAMMA && c.block == BVG_GAMMA && c.residual == BVG_ZETA;
        // experimental (BVG_WG=2|4; measured: +4 % at 2 wavefronts on the eu shape, slower on sparse graphs and at 4): off by default
        if (kExperimental && knob("BVG_WG") && !materialise && !wide && !batch && dflt && a.emit_tasks && g->skip_mode == 0 && rows_default) {
            const int w = atoi(knob("BVG_WG")); wg_nw = (w == 2 || w == 4) ? w : 0;
        }
    }
    // the big-LDS classes hold few workgroups per CU: several wavefronts per pool paid there in round 1 (BVG_WGC=2|4|8 selects them)
    int wg_class = 0;
    {
        const Codings& c = a.cod;
        const bool dflt = c.outdegree == BVG_GAMMA && c.reference == BVG_UNARY && c.block_count == BVG_GAMMA && c.block == BVG_GAMMA && c.residual == BVG_ZETA;
        if (!materialise && !wide && !batch && dflt && a.emit_tasks && g->skip_mode == 0 && rows_default) wg_class = kExperimental && knob("BVG_WGC") ? atoi(knob("BVG_WGC")) : 0;   // round 1 (8 GiB eu): 258.6 ms (0), 254.9 (2), 254.6 (4); end of round 2, after the single-wavefront kernel got the window overlay and the leaf pass (2 GiB eu15 / eu): 53.2 / 53.4 ms (0), 53.3 / 53.8 (2), 54.2 / 54.6 (4), 56.5 / 57.7 (8) -- the workgroup kernel is opt-in again
        if (wg_class != 2 && wg_class != 4 && wg_class != 8) wg_class = 0;
    }
    // The flow scan kernel as tier 0 (bvg_flow.hip): full scans, default codings, 32-bit successors, windows up to 64.  Its LDS holds
    // only the lists of the window that are really copied from, so it keeps more wavefronts resident than the row kernel.
    bool flow = false; uint32_t flow_ring = 0;
    {
        const Codings& c = a.cod;
        const bool dflt = c.outdegree == BVG_GAMMA && c.reference == BVG_UNARY && c.block_count == BVG_GAMMA && c.block == BVG_GAMMA && c.residual == BVG_ZETA;
        if (kExperimental && knob("BVG_FLOW") && atoi(knob("BVG_FLOW")) && !materialise && !wide && !batch && dflt && g->skip_mode == 0 && rows_default && sh->p.window_size <= kMaxWindow) {
            flow = true;
            flow_ring = knob("BVG_FLOW_RING") ? (uint32_t)std::min(8192, std::max(512, atoi(knob("BVG_FLOW_RING")))) : 1536u;
            const size_t per = flow_scratch_bytes_per_wave(sh->p.window_size);
            const uint32_t per_cu = (uint32_t)std::min<size_t>(20, (160 * 1024) / (flow_lds_bytes(flow_ring) + 1536 + 64));
            const uint32_t waves = 256u * std::max(1u, per_cu);
            if (g->flow_waves != waves || g->flow_ws_bytes < per * waves) {
                if (g->flow_ws) { (void)hipFree(g->flow_ws); g->flow_ws = nullptr; g->flow_ws_bytes = 0; }
                if (hipMalloc(&g->flow_ws, per * waves) != hipSuccess) { (void)hipGetLastError(); flow = false; }
                else { g->flow_ws_bytes = per * waves; g->flow_waves = waves; }
            }
        }
    }
    auto launch_rows_any = [&](const DecodeArgs& aa, uint32_t nb, hipStream_t st, bool is_class = false) {
        if (flow && !is_class && aa.work_list == g->pred.d_lists) { launch_flow_scan(aa, nb, g->flow_waves, g->flow_ws, flow_ring, st); return; }
        const int nw = is_class && wg_class ? wg_class : wg_nw;
        if (nw) launch_rows_wg_decode(aa, nb, nw, st); else launch_rows_decode(aa, nb, wide, materialise, st);
    };
    // tier 2a (bvg_giant.hip): lists / records too large for LDS, decoded by a whole workgroup each; default codings and windows <= 64
    // (anything else, and whatever it refuses, takes the generic kernel).  BVG_GIANT=0 switches it off.
    bool giant_ok = false;
    {
        const Codings& c = a.cod;
        const bool dflt = c.outdegree == BVG_GAMMA && c.reference == BVG_UNARY && c.block_count == BVG_GAMMA && c.block == BVG_GAMMA && c.residual == BVG_ZETA;
        giant_ok = dflt && rows_default && sh->p.window_size <= kMaxWindow && !(knob("BVG_GIANT") && atoi(knob("BVG_GIANT")) == 0);
    }
    uint32_t launches = 0, slow_blocks = 0, lean_blocks = 0;
    bool predicted_run = false;                        // cascade outcomes of a predicted run are remembered in g->pred
    double kernel_ms = 0;
    std::vector<uint32_t> work;
    uint32_t* d_work = nullptr;
    auto fetch_failures = [&](std::vector<uint32_t>& out) -> int {
        uint32_t nfail = 0;
        HIPCHK(hipMemcpy(&nfail, g->d_fail, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (nfail > g->fail_cap) return BVG_E_NOMEM;
        out.resize(nfail);
        if (nfail) HIPCHK(hipMemcpy(out.data(), g->d_fail + 1, nfail * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return 0;
    };
    auto timed = [&](const char* what, size_t nb, auto&& launch) -> int {
        HIPCHK(hipEventRecord(g->ev0, g->stream));
        launch();
        HIPCHK(hipEventRecord(g->ev1, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
        kernel_ms += ms;
        if (dbg_on()) fprintf(stderr, "[bvg] %s: %zu blocks, %.3f ms\n", what, nb, ms);
        return 0;
    };

    auto upload_work = [&]() -> int {
        if (d_work) { (void)hipFree(d_work); d_work = nullptr; }
        HIPCHK(hipMalloc(&d_work, work.size() * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(d_work, work.data(), work.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemsetAsync(g->d_fail, 0, sizeof(uint32_t), g->stream));
        a.work_list = d_work;
        return 0;
    };
    // ---- tier 0: every block, LDS sized for occupancy (the list pool holds one row of 64 lists + the window)
    if (nblocks && !force_slow && !force_giant) {
        if (stream) {                                       // list ring: power of two
            uint64_t want = (uint64_t)(avg * 72.0), cap = 2048;
            while (cap * 2 <= want && cap < (wide ? 8192u : 16384u)) cap *= 2;
            if (knob("BVG_POOL")) cap = strtoull(knob("BVG_POOL"), nullptr, 10);
            a.lds_pool_elems = (uint32_t)cap; a.lds_scr_elems = 0;
        } else {
            const bool task = a.emit_tasks != 0;                          // task emission parks the row's residuals beside the lists
            uint64_t pool = ((uint64_t)(avg * (task ? 52.0 : 48.0)) + 255) & ~255ull;   // ~a row of lists (rows shrink when they do not fit)
            pool = std::min<uint64_t>(std::max<uint64_t>(pool, 1024), wide ? 4096 : 8192);
            if (wg_nw) {
                // workgroups per CU are bounded by registers (wavefronts per SIMD): give each the LDS share of that count
                uint64_t wgs = wg_nw == 4 ? 5 : 8;
                if (knob("BVG_WG_BLOCKS")) wgs = std::max<uint64_t>(1, strtoull(knob("BVG_WG_BLOCKS"), nullptr, 10));
                const uint64_t share = ((160 * 1024) / wgs) & ~255ull, fixed = (uint64_t)a.lds_stage_words * 4 + rows_wg_static_lds(wg_nw) + 256;
                const uint64_t fit = share > fixed ? ((share - fixed) / esz) * 8 / 9 : 1024;      // pool + pool/8 of scratch
                pool = std::min<uint64_t>(std::max<uint64_t>(pool, fit & ~63ull), 12288);
                pool = std::max<uint64_t>(pool, 1024);
            } else if (task) {
                // resident waves per CU step down with the LDS footprint: take every byte of the step the pool lands on
                const uint64_t lds_cu = 160 * 1024, fixed = 1536 + 64;   // static arrays (+ slack); the task variant keeps the stream window INSIDE the pool
                auto foot = [&](uint64_t pe) { return ((pe + std::max<uint64_t>(256, pe / 8)) * esz + fixed + 127) & ~127ull; };
                uint64_t waves = std::max<uint64_t>(1, lds_cu / foot(pool));
                // Two wavefronts per SIMD (8 per CU) is the step that pays on dense graphs: below it the CU's SIMDs sit idle behind
                // LDS latency, and a row that shrinks to ~40 lists costs less than the lost wavefronts (eu15 shape, 4 GiB: 90.0 G
                // edges/s at 6 per CU with 54 lists per row, 98.7 G at 8 per CU with 43; profiles/r02/occ_sweep15.sh).
                // Resident wavefronts per CU are what this kernel's throughput follows (linear from 1 to 8, profiles/r02/ldspad.sh), as
                // long as a row still holds enough lists to fill its lock-step passes: take the largest EVEN count (odd ones load
                // the four SIMDs unevenly: 9 and 11 measured below 8 and 10) whose pool holds ~48 average lists; dense graphs end at
                // 8-10, sparse ones at the 16 the registers allow (profiles/r02: eu 10 per CU 118.8 G edges/s vs 8: 117.3, 9: 113.8;
                // eu15 8: 121.6, 9: 111.0, 10: 111.3).
                if (!knob("BVG_STAGE")) {
                    for (uint64_t w : {20ull, 16ull, 12ull, 10ull, 8ull, 6ull, 4ull}) {
                        uint64_t pw = wide ? 4096 : 8192;
                        while (pw > 1024 && lds_cu / foot(pw) < w) pw -= 32;
                        if (lds_cu / foot(pw) >= w && ((double)pw >= 48.0 * avg || w == 4)) { pool = pw; waves = lds_cu / foot(pw); a.lds_stage_words = std::min<uint32_t>(a.lds_stage_words, 512); break; }
                    }
                }
                if (knob("BVG_WAVES")) {                                 // experiments: aim at this many resident wavefronts per CU
                    const uint64_t w = std::max<uint64_t>(1, strtoull(knob("BVG_WAVES"), nullptr, 10));
                    uint64_t pw = wide ? 4096 : 8192;
                    while (pw > 1024 && lds_cu / foot(pw) < w) pw -= 64;
                    pool = pw; waves = lds_cu / foot(pw);
                }
                while (pool + 32 <= (wide ? 4096u : 8192u) && lds_cu / foot(pool + 32) == waves) pool += 32;
            }
            if (knob("BVG_POOL")) pool = std::min<uint64_t>(std::max<uint64_t>(strtoull(knob("BVG_POOL"), nullptr, 10), 256), wide ? 6144 : 12288);
            a.lds_pool_elems = (uint32_t)pool; a.lds_scr_elems = (uint32_t)std::max<uint64_t>(256, pool / 8);
        }
        if (batch) {                                        // one block per request: the even entries of the per-call plan
            work.resize(nblocks); for (uint32_t i = 0; i < nblocks; i++) work[i] = 2 * i;
            r = upload_work(); if (r) return r;
            work.clear();
        }
        const uint32_t max_pool = 12288;                              // (elements, whatever their width: 127 KB of LDS for the largest class of the 64-bit row kernel -- it is what VALIDATES such a block for the scan kernel, whose lists are 32-bit on every graph)
        const uint32_t classes[4] = {max_pool / 6, max_pool / 3, (max_pool * 2) / 3, max_pool};
        const uint32_t lclasses[4] = {2048, 4096, 8192, 12288};               // the lean scan kernel's lists are 32-bit on every graph (block-relative ids beyond 2^32 nodes)
        const bool predict = !batch && !stream && !legacy && pl.h_maxd.size() == pl.nblk && !knob("BVG_NOPREDICT");
        // The lean scan kernel (bvg_scan.hip) takes the blocks that the index-building pass has validated: scans with 32-bit
        // successors and the default codings, index present.  BVG_SCANK=0 keeps every block on the row kernel (tests, A/B runs).
        bool fast_ok = false; uint32_t lean_waves = 0;
        // The flat scan kernel (experimental/bvg_flat.hip, round 5: bit-exact, slower -- DESIGN.md) takes what the lean scan kernel takes, for scans (not materialising
        // calls) of graphs whose ids fit 32 bits, in the experimental build with BVG_FLAT=1; BVG_FLAT_RECS = records per super-row (64 ... 256).
        const bool flat_on = kExperimental && !materialise && !wide && knob("BVG_FLAT") && atoi(knob("BVG_FLAT")) == 1;
        uint32_t flat_recs = avg <= 16.0 ? 128u : 64u;
        if (knob("BVG_FLAT_RECS")) flat_recs = std::min(256u, std::max(64u, (unsigned)atoi(knob("BVG_FLAT_RECS")) & ~63u));
        const size_t lean_static = flat_on ? flat_table_bytes(flat_recs, sh->p.window_size) : scan_static_lds();
        auto launch_lean = [&](DecodeArgs& al, uint32_t nb, bool many_waves, hipStream_t st) {
            if (flat_on) { al.flat_recs = flat_recs; launch_flat_decode(al, nb, many_waves, st); }
            else launch_scan_decode(al, nb, wide, many_waves, materialise, st);
        };
        DecodeArgs af = a;
        {
            const Codings& c = a.cod;
            const bool dflt = c.outdegree == BVG_GAMMA && c.reference == BVG_UNARY && c.block_count == BVG_GAMMA && c.block == BVG_GAMMA && c.residual == BVG_ZETA;
            // (a materialising call takes it on dense graphs only: below ~16 arcs per node the row kernel's pipelined loop is the faster way to
            //  build every list -- web shape 71.9 vs 60.5 G edges/s, eu shape 93.6 vs 165.9: profiles/r04_mat_first.txt)
            fast_ok = predict && !(materialise && avg < 16.0 && !knob("BVG_MAT_LEAN")) && dflt && a.emit_tasks && g->skip_mode == 0 && rows_default && a.skip_first && a.skip_fmt && skx &&
                      skx->h_fmt.size() == pl.nblk && !wg_nw && !flow && !(knob("BVG_SCANK") && atoi(knob("BVG_SCANK")) == 0);
            if (fast_ok) {
                // LDS per wavefront: pool (stored lists + their parked residuals + the window) + scratch (copy blocks, intervals, run
                // queue) + static arrays.  Resident wavefronts per CU step down with it; take the largest even count whose pool
                // still holds a row's worth of lists (leaves take no pool: about half the row kernel's need).
                // The window: 512 dwords at up to 14 wavefronts, 384 at 16 (profiles/r03_ab_uni.txt; a super-row = the records that fit it, up to 64).
                // Lists, parked residuals and the super-row's copy blocks / intervals share pool + scratch (bvg_scan.hip): about half the
                // scratch is free for lists on average, and counts as such here.
                const uint64_t scrw = knob("BVG_SCAN_SCR") ? strtoull(knob("BVG_SCAN_SCR"), nullptr, 10) : 448;
                const uint64_t lds_cu = 160 * 1024;
                auto stage_of = [&](uint64_t w) -> uint32_t {
                    return knob("BVG_SCAN_STAGE") ? (uint32_t)std::min(2048, std::max(128, atoi(knob("BVG_SCAN_STAGE")) & ~3)) : std::min<uint32_t>(a.lds_stage_words, w >= 16 ? 384 : 512);
                };
                auto foot = [&](uint64_t pe, uint64_t w) { return (pe * 4 + lean_static + 64 + (uint64_t)stage_of(w) * 4 + scrw * 4 + 127) & ~127ull; };
                const double lists = knob("BVG_SCAN_LISTS") ? atof(knob("BVG_SCAN_LISTS")) : 20.0;   // window lists + a sub-row's stored lists and parked residuals
                uint64_t pool = 1024, waves = 4;
                const uint64_t wforce = knob("BVG_SCAN_WAVES") ? strtoull(knob("BVG_SCAN_WAVES"), nullptr, 10) : 0;
                for (uint64_t w : {24ull, 20ull, 16ull, 14ull, 12ull, 10ull, 8ull, 6ull, 4ull}) {
                    if (w > 16 && wforce != w && !(w == 24 && avg <= 16.0 && sh->p.window_size > 0 && !wforce && !materialise)) continue;   // more than 16: the 85-VGPR instantiation, sparse graphs with references only (web shape: +7 %; eu15: -11 % at 20; w0, all residuals: -6 %)
                    if (wforce && w != wforce && w != 4) continue;
                    uint64_t pw = 8192;
                    while (pw > 512 && lds_cu / foot(pw, w) < w) pw -= 32;
                    if (lds_cu / foot(pw, w) >= w && ((double)(pw + scrw / 2) >= lists * avg || w == 4 || wforce)) { pool = pw; waves = w; break; }
                }
                const uint32_t stagew = stage_of(waves); lean_waves = (uint32_t)waves;
                while (pool + 32 <= 8192 && lds_cu / foot(pool + 32, waves) >= waves) pool += 32;
                if (knob("BVG_SCAN_POOL")) pool = std::min<uint64_t>(std::max<uint64_t>(strtoull(knob("BVG_SCAN_POOL"), nullptr, 10), 512), 12288);
                af.lds_pool_elems = (uint32_t)pool; af.lds_scr_elems = (uint32_t)scrw;
                af.lds_stage_words = stagew;
                if (dbg_on()) fprintf(stderr, "[bvg] %s: pool %u + scratch %u elements, window %u dwords, %llu wavefronts per CU\n", flat_on ? "flat kernel" : "scan kernel", af.lds_pool_elems, af.lds_scr_elems, af.lds_stage_words, (unsigned long long)waves);
            }
        }
        if (predict) {
            // blocks sorted into {tier 0, four LDS size classes, giants} by the largest list they hold
            bvg_graph::Pred& pd = g->pred;
            const uint32_t pool0 = a.lds_pool_elems;
            const uint32_t pmode = (materialise ? 1u : 0u) | (a.emit_tasks ? 2u : 0u) | (a.skip_first ? 4u : 0u) | (wide ? 8u : 0u) | (flow ? 16u : 0u) | (fast_ok ? 32u : 0u) | (fast_ok && flat_on ? 64u : 0u) | (fast_ok ? (af.lds_pool_elems << 8) : 0u);
            const uint64_t cap0 = flow ? 6144 : pool0;                          // the flow kernel keeps long lists in its scratch area
            const uint64_t sgen = (a.skip_first && skx) ? skx->gen : 0;         // the snapshot the marks / entry layouts come from: another one, another split
            const bool rekey = pd.plan_version != pl.version || pd.skip_gen != sgen || pd.lo != lo || pd.n != nblocks || pd.pool0 != pool0 || pd.mode != pmode || !pd.d_lists;
            // what the cascade taught about a block is kept per block of the PLAN, so a scan of another node range (a shard, an
            // iterator batch, the bench's verification of single tiles) does not throw it away
            if (pd.learned.size() != pl.nblk || pd.learned_version != pl.version || pd.learned_gen != sgen || pd.learned_pool0 != pool0 || pd.learned_mode != pmode) {
                pd.learned.assign(pl.nblk, 0); pd.leanfail.assign(pl.nblk, 0); pd.learned_version = pl.version; pd.learned_gen = sgen; pd.learned_pool0 = pool0; pd.learned_mode = pmode;
            }
            if (rekey) pd.dirty = false;
            if (rekey || pd.dirty) {
                std::vector<uint32_t> L[12];                                     // tier 0, four LDS classes, giants (5), the generic kernel (6); 7..11: tier 0 and the classes of the lean scan kernel
                uint64_t gneed = 0, gnodes = 0, glong = 0;
                const double cadmit = knob("BVG_CADMIT") ? atof(knob("BVG_CADMIT")) : 0.75;   // the same optimism for the lean classes (a block that fails its class is learned upward): +0.8 % on the default workload (profiles/r05_ab_cadmit.txt)
                const double admit = knob("BVG_ADMIT") ? atof(knob("BVG_ADMIT")) : 0.3;   // share of a block's worst "list + window" that tier 0 of the scan kernel must hold
                for (uint32_t i = 0; i < nblocks; i++) {
                    const uint64_t md = pl.h_maxd[lo + i] & 0x7FFFFFFFu;       // worst "list + window" of the block
                    const bool long_record = (pl.h_maxd[lo + i] >> 31) != 0;
                    const uint64_t need = md + md / 8 + 64;
                    const bool fastb = fast_ok && skx->h_fmt[lo + i] == 1 && pd.leanfail[lo + i] < 2;   // (a block the lean kernel failed twice -- first for its pool, then in the class it was sent to -- stays on the row kernel)
                    int c;
                    if (long_record) c = 5;
                    else if (fastb ? ((uint64_t)((double)md * admit) + 64 <= af.lds_pool_elems + af.lds_scr_elems / 2) : need <= cap0) c = 0;   // (the lean kernel stores only the lists that are copied from: optimistic, the cascade teaches the rest)
                    else { c = 1; const uint64_t cneed = fastb ? (uint64_t)((double)need * cadmit) : need; while (c < 5 && (fastb ? lclasses : classes)[c - 1] < cneed) c++; }
                    const int lrn = pd.learned[lo + i];                        // learned from an earlier scan's cascade
                    if (lrn > c) { c = lrn; if (c >= 5 && gneed < 65536) gneed = 65536; }
                    if (c == 5 && !giant_ok) c = 6;
                    if (c >= 5 && need > gneed) gneed = need;
                    if (c >= 5) { gnodes += pl.h_first[lo + i + 1] - pl.h_first[lo + i]; if (long_record) glong++; }
                    L[(fastb && c <= 4) ? 7 + c : c].push_back(lo + i);
                }
                if (dbg_on() && L[5].size() + L[6].size()) fprintf(stderr, "[bvg] giant blocks: %zu (%llu of them for a record longer than the window), %llu nodes in them\n", L[5].size() + L[6].size(), (unsigned long long)glong, (unsigned long long)gnodes);
                pd.dirty = false; pd.mode = pmode;
                if (pd.d_lists) { (void)hipFree(pd.d_lists); pd.d_lists = nullptr; }
                HIPCHK(hipMalloc(&pd.d_lists, (size_t)nblocks * sizeof(uint32_t)));
                size_t off = 0;
                for (int c = 0; c < 12; c++) {
                    pd.count[c] = (uint32_t)L[c].size();
                    if (!L[c].empty()) HIPCHK(hipMemcpy(pd.d_lists + off, L[c].data(), L[c].size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                    off += L[c].size();
                }
                pd.plan_version = pl.version; pd.skip_gen = sgen; pd.lo = lo; pd.n = nblocks; pd.pool0 = pool0; pd.giant_need = gneed;
            }
            // giants: global-memory pools sized to the largest list, allocated before anything is launched
            uint64_t gpool_elems = 0, gscr_elems = 0; uint32_t gbatch = 0; bool use_slots = false;
            const uint32_t ngiant = pd.count[5] + pd.count[6];
            if (ngiant) {
                // (the giant kernel parks the residuals of the list it decodes in the same area: twice the worst list + window)
                gpool_elems = 1ull << 16; while (gpool_elems < 2 * pd.giant_need + pd.giant_need / 4) gpool_elems <<= 1;
                // Work areas: as many SLOTS as giant workgroups can be resident at once (2 per CU: bvg_giant.hip) and half as many again, whatever the
                // number of giant blocks -- the kernel takes a free slot when a workgroup starts (DecodeArgs::gslots).  Round 3 sized one area per block of
                // a batch of 8 192 (up to 1/8 of the free memory: 26-31 GB on the default workload, per handle).  All giants go in ONE launch.
                gscr_elems = gpool_elems / 2; gbatch = std::min<uint32_t>(giant_slots(), ngiant);
                if (knob("BVG_GBATCH")) gbatch = (uint32_t)std::max(1, atoi(knob("BVG_GBATCH")));   // (experiments: batched launches, one area per block of a batch)
                use_slots = !knob("BVG_GBATCH");
                {
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                        const uint64_t per = (gpool_elems + gscr_elems) * esz, room = ((uint64_t)free_b + g->giant_ws_bytes) / 4;
                        const uint64_t fit = room / std::max<uint64_t>(per, 1);
                        if (fit < gbatch) { gbatch = (uint32_t)std::max<uint64_t>(fit, 1); if (gbatch < std::min<uint32_t>(kGiantResident, ngiant)) use_slots = false; }   // too few slots for every resident workgroup: batches again
                    }
                }
                const uint64_t bytes = (uint64_t)gbatch * (gpool_elems + gscr_elems) * esz;
                if (bytes > g->giant_ws_bytes) {
                    if (g->giant_ws) { (void)hipFree(g->giant_ws); g->giant_ws = nullptr; g->giant_ws_bytes = 0; }
                    if (hipMalloc(&g->giant_ws, bytes) == hipSuccess) g->giant_ws_bytes = bytes; else gbatch = 0;   // fall back to the cascade
                }
                if (use_slots && gbatch) {
                    if (!g->d_gslots && hipMalloc(&g->d_gslots, 8192 * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); g->d_gslots = nullptr; use_slots = false; gbatch = std::min<uint32_t>(gbatch, 256u); }
                    if (use_slots) HIPCHK(hipMemsetAsync(g->d_gslots, 0, 8192 * sizeof(uint32_t), g->stream));   // (ordered before the side streams by ev0 below)
                }
            }
            HIPCHK(hipEventRecord(g->ev0, g->stream));
            for (int i = 0; i < bvg_graph::kSide; i++) HIPCHK(hipStreamWaitEvent(g->side[i], g->ev0, 0));
            DecodeArgs a0 = a; a0.work_list = pd.d_lists;                      // tier 0 on the main stream
            size_t offc[12]; { size_t o = 0; for (int c = 0; c < 12; c++) { offc[c] = o; o += pd.count[c]; } }
            const bool tier0_first = knob("BVG_ORDER") && atoi(knob("BVG_ORDER")) == 1;
            bool t0_waits = false;
            // side streams: [0] the giants and, behind them, the smallest class (short); [1..3] one per larger LDS class, so that every
            // class starts with the main launch and overlaps it.  (One stream per class and one for the giants made six streams: the
            // largest class then started only when the last giant batch had finished -- streams share hardware queues -- and ended 11 ms
            // after everything else at full size; three side streams were 11 % slower, profiles/r03_ab_smap.txt.)
            const int smap = knob("BVG_SIDE2") ? atoi(knob("BVG_SIDE2")) : 0;
            auto side_of = [&](int c) {
                if (smap == 1) return g->side[c >= 4 ? 1 : c == 3 ? 0 : 2];
                if (smap == 2) return g->side[c];
                return g->side[c == 1 ? 0 : c - 1];
            };
            const bool serial = knob("BVG_SERIAL") != nullptr;               // experiments: every launch alone on the chip (its own duration in a kernel trace)
            auto alone = [&](hipStream_t st) { if (serial) (void)hipStreamSynchronize(st); };
            if (tier0_first && pd.count[0]) { launch_rows_any(a0, pd.count[0], g->stream); launches++; }
            if (tier0_first && pd.count[7]) { DecodeArgs a7 = af; a7.work_list = pd.d_lists + offc[7]; launch_lean(a7, pd.count[7], lean_waves > 20 || (lean_waves > 16 && !knob("BVG_SCAN_OCC")), g->stream); launches++; alone(g->stream); }
            if (ngiant && gbatch) {                                            // giants first: they are the critical path
                DecodeArgs ag = a; ag.gpool = g->giant_ws; ag.gpool_elems = gpool_elems;
                if (ag.skip_mode == 3) ag.skip_mode = 2;                              // (the giant kernel fills its own entries, in its own format, while it validates)
                ag.gscr = (char*)g->giant_ws + (size_t)gbatch * gpool_elems * esz; ag.gscr_elems = gscr_elems; ag.lds_stage_words = 1024;
                for (int c = 5; c <= 6; c++) {
                    const bool slots = use_slots && c == 5;                       // (the generic kernel keeps one area per block of a batch)
                    ag.gslots = slots ? g->d_gslots : nullptr; ag.gnslots = slots ? gbatch : 0u;
                    const uint32_t step = slots ? std::max<uint32_t>(pd.count[c], 1u) : gbatch;
                    for (uint32_t o2 = 0; o2 < pd.count[c]; o2 += step) {
                        ag.work_list = pd.d_lists + offc[c] + o2;
                        const uint32_t nb = std::min<uint32_t>(step, pd.count[c] - o2);
                        if (c == 5) launch_giant_decode(ag, nb, wide, materialise, g->side[0]);
                        else launch_decode(ag, nb, wide, materialise, true, g->side[0]);
                        launches++; alone(g->side[0]);
                    }
                }
                // (experiment, BVG_T0WAIT=1: tier 0 starts when the giants are done.  With all giants in one launch they trickle through the whole scan beside tier 0 -- a giant
                // workgroup needs 16 wave slots of ONE CU at once -- and end ~30 ms after it, profiles/r04_eu15_scan_timeline.txt; holding tier 0 back by the giants' ~30 ms
                // ends the scan on tier 0 instead and takes exactly as long: 366.0 vs 366.5 ms, profiles/r04_ab_t0wait.txt.  The launches are work-conserving.)
                if (knob("BVG_T0WAIT") && atoi(knob("BVG_T0WAIT")) == 1 && !tier0_first) { HIPCHK(hipEventRecord(g->side_ev[0], g->side[0])); t0_waits = true; }
            }
            for (int c = 4; c >= 1; c--) {                                     // LDS size classes, largest first
                if (!pd.count[c]) continue;
                DecodeArgs ac = a; ac.work_list = pd.d_lists + offc[c];
                ac.lds_pool_elems = classes[c - 1]; ac.lds_scr_elems = std::max<uint32_t>(1024, classes[c - 1] / 4); ac.lds_stage_words = 1024;
                launch_rows_any(ac, pd.count[c], side_of(c), true); alone(side_of(c));
                launches++;
            }
            for (int c = 4; c >= 1; c--) {                                     // the same classes of the lean scan kernel
                if (!pd.count[7 + c]) continue;
                DecodeArgs ac = af; ac.work_list = pd.d_lists + offc[7 + c];
                // LDS geometry of the lean classes (round 4): windows of 256 / 384 / 512 / 768 dwords and scratch areas of 384 / 512 / 1 024 / 2 048 elements instead of
                // 1 024 dwords and >= 1 024 elements throughout -- resident wavefronts per CU 9 -> 14 / 6 -> 8 / 3 -> 4 in the three populated classes; alone on the chip
                // they take 53.3 instead of 63.3 ms, in the concurrent schedule the scan gains 1.9 % (profiles/r04_serial_classes.txt, r04_ab_prio.txt); the few
                // blocks whose longest record no longer fits the window fail over to the row kernel's classes (1 821 of 774 k)
                static const uint32_t cstage[4] = {256, 384, 512, 768}, cscr[4] = {384, 512, 1024, 2048};
                ac.lds_pool_elems = lclasses[c - 1]; ac.lds_scr_elems = cscr[c - 1]; ac.lds_stage_words = cstage[c - 1];
                if (knob("BVG_CLASS_STAGE")) { unsigned v[4] = {1024, 1024, 1024, 1024}; sscanf(knob("BVG_CLASS_STAGE"), "%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3]); ac.lds_stage_words = std::min(2048u, std::max(128u, v[c - 1] & ~3u)); }   // experiments: the classes' LDS geometry
                if (knob("BVG_CLASS_SCR")) { unsigned v[4] = {1024, 1024, 2048, 3072}; sscanf(knob("BVG_CLASS_SCR"), "%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3]); ac.lds_scr_elems = std::min(8192u, std::max(128u, v[c - 1])); }
                launch_lean(ac, pd.count[7 + c], false, side_of(c)); alone(side_of(c));
                launches++;
            }
            if (t0_waits) HIPCHK(hipStreamWaitEvent(g->stream, g->side_ev[0], 0));
            if (!tier0_first && pd.count[7]) { DecodeArgs a7 = af; a7.work_list = pd.d_lists + offc[7]; launch_lean(a7, pd.count[7], lean_waves > 20 || (lean_waves > 16 && !knob("BVG_SCAN_OCC")), g->stream); launches++; alone(g->stream); }
            if (!tier0_first && pd.count[0]) { launch_rows_any(a0, pd.count[0], g->stream); launches++; }
            for (int i = 0; i < bvg_graph::kSide; i++) { HIPCHK(hipEventRecord(g->side_ev[i], g->side[i])); HIPCHK(hipStreamWaitEvent(g->stream, g->side_ev[i], 0)); }
            HIPCHK(hipEventRecord(g->ev1, g->stream));
            HIPCHK(hipStreamSynchronize(g->stream));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
            kernel_ms += ms;
            if (dbg_on()) fprintf(stderr, "[bvg] tiers concurrent: scan kernel %u + %u/%u/%u/%u LDS-class, row kernel %u + %u/%u/%u/%u LDS-class, %u giant + %u generic blocks, %.3f ms\n",
                                             pd.count[7], pd.count[8], pd.count[9], pd.count[10], pd.count[11], pd.count[0], pd.count[1], pd.count[2], pd.count[3], pd.count[4], pd.count[5], pd.count[6], ms);
            slow_blocks = nblocks - pd.count[0] - pd.count[7];
            lean_blocks = pd.count[7] + pd.count[8] + pd.count[9] + pd.count[10] + pd.count[11];
            predicted_run = true;
            if (ngiant && !gbatch) {                                           // could not get the giant workspace: leave them to the cascade
                std::vector<uint32_t> gl(ngiant);
                HIPCHK(hipMemcpy(gl.data(), pd.d_lists + offc[5], gl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                r = fetch_failures(work); if (r) return r;
                work.insert(work.end(), gl.begin(), gl.end());
            } else { r = fetch_failures(work); if (r) return r; }
            if (fast_ok) for (uint32_t id : work) if (id < pd.leanfail.size() && skx->h_fmt[id] == 1 && pd.leanfail[id] < 2) { pd.leanfail[id]++; pd.dirty = true; }
            slow_blocks += (uint32_t)work.size();                              // blocks the prediction missed: re-run by the cascade below
        } else {
#ifdef BVG_EXPERIMENTAL
        r = timed("tier0 (LDS)", nblocks, [&] { if (stream) launch_stream_decode(a, nblocks, wide, materialise, g->stream);
                                              else if (legacy) launch_decode(a, nblocks, wide, materialise, false, g->stream);
                                              else launch_rows_any(a, nblocks, g->stream); });
#else
        r = timed("tier0 (LDS)", nblocks, [&] { launch_rows_any(a, nblocks, g->stream); });
#endif
        if (r) return r;
        launches++;
        r = fetch_failures(work); if (r) return r;
        slow_blocks = (uint32_t)work.size();
        }
    } else if (force_slow || force_giant) { work.resize(nblocks); for (uint32_t i = 0; i < nblocks; i++) work[i] = batch ? 2 * i : lo + i; slow_blocks = nblocks; }

    // ---- tier 1: the few blocks holding a list that overflowed the small pool, re-run with a pool sized to
    //      what each block reported it needs (size classes keep as many waves resident as possible)
    if (!work.empty() && !force_slow && !force_giant) {
        std::vector<uint32_t> need(work.size());
        HIPCHK(hipMemcpy(need.data(), g->d_fail + 1 + g->fail_cap, work.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t max_pool = 12288;                              // (elements, whatever their width: 127 KB of LDS for the largest class of the 64-bit row kernel -- it is what VALIDATES such a block for the scan kernel, whose lists are 32-bit on every graph)
        const uint32_t classes[4] = {max_pool / 6, max_pool / 3, (max_pool * 2) / 3, max_pool};
        std::vector<uint32_t> bins[4], rest;
        if (dbg_on()) { size_t h[8] = {0}; for (uint32_t nd : need) h[nd >= 0xFFFFFFF0u ? (nd & 7) : 0]++; fprintf(stderr, "[bvg] failures: pool %zu, window %zu, huge %zu, blocks-scratch %zu, intervals-scratch %zu, code %zu, other %zu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[7]); }
        for (size_t i = 0; i < work.size(); i++) {
            int c = 3;
            if (!stream && !legacy && need[i] < 0xFFFFFFF0u) { c = 0; while (c < 3 && classes[c] < need[i]) c++; if (classes[c] < need[i]) c = -1; }
            if (c < 0) rest.push_back(work[i]); else bins[c].push_back(work[i]);
        }
        for (int c = 0; c < 4; c++) {
            if (bins[c].empty()) continue;
            work.swap(bins[c]);
            r = upload_work(); if (r) return r;
            a.lds_pool_elems = classes[c]; a.lds_scr_elems = std::max<uint32_t>(1024, classes[c] / 4); a.lds_stage_words = 1024;
            const uint32_t nb = (uint32_t)work.size();
            r = timed("tier1 (big LDS)", nb, [&] { if (legacy) launch_decode(a, nb, wide, materialise, false, g->stream); else launch_rows_any(a, nb, g->stream, true); });
            if (r) return r;
            launches++;
            std::vector<uint32_t> again;
            r = fetch_failures(again); if (r) return r;
            // what a class fails is tried in the next larger one (the need a block reported may come from another kernel's footprint)
            if (c < 3) bins[c + 1].insert(bins[c + 1].end(), again.begin(), again.end()); else rest.insert(rest.end(), again.begin(), again.end());
            if (predicted_run) {                               // remember where the survivors of this class fit
                bvg_graph::Pred& pd = g->pred;
                std::sort(again.begin(), again.end());
                for (uint32_t id : work)
                    if (id < pd.learned.size() && !std::binary_search(again.begin(), again.end(), id)) { pd.learned[id] = (uint8_t)(c + 1); pd.dirty = true; }
            }
        }
        work.swap(rest);
        if (predicted_run) { bvg_graph::Pred& pd = g->pred; for (uint32_t id : work) if (id < pd.learned.size()) { pd.learned[id] = 5; pd.dirty = true; } }
    }
    // ---- tier 2a / 2: per-workgroup areas in global memory (kept in the handle), grown until every remaining block fits.  First the
    //      giant kernel (a workgroup per list); what it refuses (overlapping streams, contradictory counts) goes to the generic kernel.
    auto run_global_tier = [&](bool giant, std::vector<uint32_t>& refused) -> int {
        uint64_t pool_elems = 1ull << 20;
        while (!work.empty()) {
            uint64_t scr_elems = pool_elems / 2;
            uint64_t per_wg = (pool_elems + scr_elems) * esz;
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            uint32_t batch = (uint32_t)std::min<uint64_t>({(uint64_t)work.size(), std::max<uint64_t>(1, ((free_b + g->slow_ws_bytes) / 2) / per_wg), 1024});
            if ((uint64_t)batch * per_wg > g->slow_ws_bytes) {
                if (g->slow_ws) { (void)hipFree(g->slow_ws); g->slow_ws = nullptr; g->slow_ws_bytes = 0; }
                if (hipMalloc(&g->slow_ws, (size_t)batch * per_wg) != hipSuccess) { if (d_work) (void)hipFree(d_work); d_work = nullptr; return BVG_E_NOMEM; }
                g->slow_ws_bytes = (uint64_t)batch * per_wg;
            }
            int r2 = upload_work(); if (r2) return r2;
            a.gpool = g->slow_ws; a.gpool_elems = pool_elems;
            a.gscr = (char*)g->slow_ws + (size_t)batch * pool_elems * esz; a.gscr_elems = scr_elems;
            a.lds_stage_words = 1024;
            const size_t nwork = work.size();
            r2 = timed(giant ? "tier2a (giant)" : "tier2 (generic)", nwork, [&] {
                for (size_t off = 0; off < nwork; off += batch) {
                    uint32_t nb = (uint32_t)std::min<size_t>(batch, nwork - off);
                    a.work_list = d_work + off;
                    if (giant) { DecodeArgs ag2 = a; if (ag2.skip_mode == 3) ag2.skip_mode = 2; launch_giant_decode(ag2, nb, wide, materialise, g->stream); } else launch_decode(a, nb, wide, materialise, true, g->stream);
                    launches++;
                }
            });
            if (r2) return r2;
            r2 = fetch_failures(work); if (r2) return r2;
            if (giant && !work.empty()) {                                     // only "the area is too small" is worth another round
                std::vector<uint32_t> need(work.size()), again;
                HIPCHK(hipMemcpy(need.data(), g->d_fail + 1 + g->fail_cap, work.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < work.size(); i++) (need[i] == 0xFFFFFFF2u ? again : refused).push_back(work[i]);
                work.swap(again);
            }
            if (!work.empty()) {
                if (pool_elems >= (1ull << 34)) { if (giant) { refused.insert(refused.end(), work.begin(), work.end()); work.clear(); break; } if (d_work) (void)hipFree(d_work); d_work = nullptr; return BVG_E_NOMEM; }
                pool_elems *= 8;
            }
        }
        return 0;
    };
    if (!work.empty() && giant_ok && !force_slow) {
        std::vector<uint32_t> refused;
        r = run_global_tier(true, refused); if (r) return r;
        work.swap(refused);
        if (predicted_run) { bvg_graph::Pred& pd = g->pred; for (uint32_t id : work) if (id < pd.learned.size()) { pd.learned[id] = 6; pd.dirty = true; } }
    }
    { std::vector<uint32_t> none; r = run_global_tier(false, none); if (r) return r; }
    if (d_work) (void)hipFree(d_work);

    unsigned long long acc[32];
    launch_reduce_acc(g->d_acc, kAccStripes, g->stream);
    HIPCHK(hipMemcpyAsync(acc, g->d_acc, sizeof acc, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    if (a.dbg & 64u) fprintf(stderr, "[bvg] counters: position steps %llu, position passes %llu, extras passes %llu, rows %llu, position tasks %llu, leaf steps %llu, leaf passes %llu\n", acc[4], acc[5], acc[8], acc[6], acc[7], acc[22], acc[23]);
    if ((a.dbg & 64u) && lean_blocks) fprintf(stderr, "[bvg] scan kernel rows: %llu super-rows, %llu sub-rows, %llu nodes in them\n", acc[5], acc[6], acc[7]);
    if ((a.dbg & 64u) && lean_blocks && knob("BVG_FLAT_PROF"))          // `make flatprof` (-DBVG_FLAT_PROF): the flat kernel's section cycles and work counts (bvg_flat.hip)
        fprintf(stderr, "[bvg] flat kernel wave-cycles (M): super-row set-up %.0f, headers %.0f, peek/marks %.0f, sizing+stages %.0f, residual set-up %.0f, residual steps %.0f, Z1 %.0f, item set-up %.0f, chunks %.0f, compaction %.0f | "
                "super-rows %llu sub-rows %llu records %llu | residual passes %llu steps %llu | Z1 passes %llu | item passes %llu chunk passes %llu chunk steps %llu\n",
                acc[9] / 1e6, acc[10] / 1e6, acc[11] / 1e6, acc[12] / 1e6, acc[13] / 1e6, acc[14] / 1e6, acc[15] / 1e6, acc[16] / 1e6, acc[17] / 1e6, acc[18] / 1e6,
                acc[19], acc[20], acc[21], acc[22], acc[23], acc[24], acc[25], acc[26], acc[27]);
#ifndef BVG_PROF_WORK
    if ((a.dbg & 64u) && acc[14]) {                         // only the -DBVG_PROF build fills these
        fprintf(stderr, "[bvg] wave-cycles (M): phase1 %.0f, row prep %.0f, level prep %.0f, task set-up %.0f, seeks %.0f, merge loop %.0f\n", acc[14] / 1e6, acc[9] / 1e6, acc[10] / 1e6, acc[11] / 1e6, acc[12] / 1e6, acc[13] / 1e6);
        fprintf(stderr, "[bvg] phase 1 split (M): row set-up %.0f, headers %.0f, pool sizing %.0f, residuals %.0f; leaf pass %.0f (loop %.0f)\n", acc[15] / 1e6, acc[16] / 1e6, acc[17] / 1e6, acc[18] / 1e6, acc[20] / 1e6, acc[21] / 1e6);
    }
#endif
#ifndef BVG_PROF_WORK
    if ((a.dbg & 64u) && acc[14])
        fprintf(stderr, "[bvg] scan kernel, more wave-cycles (M): compaction %.0f, window staging %.0f, residual task set-up %.0f, stored-list marking %.0f\n", acc[24] / 1e6, acc[25] / 1e6, acc[26] / 1e6, acc[27] / 1e6);
#else
    if ((a.dbg & 64u) && (acc[24] | acc[25] | acc[26] | acc[27]))   // only the -DBVG_PROF -DBVG_PROF_WORK build (`make work`): the slots above hold counts, not cycles
        fprintf(stderr, "[bvg] scan kernel work: levels %llu | Z1 passes %llu tasks %llu | Z2 passes %llu tasks %llu steps %llu positions %llu | residual task passes %llu steps %llu residuals %llu, "
                "lane-per-node steps %llu residuals %llu | leaf item passes %llu chunk passes %llu steps(x4) %llu elements %llu\n",
                acc[9], acc[12], acc[25], acc[10], acc[11], acc[13], acc[14], acc[15], acc[16], acc[17], acc[18], acc[26], acc[27], acc[20], acc[21], acc[24]);
    if ((a.dbg & 64u) && (acc[28] | acc[29]))
        fprintf(stderr, "[bvg] scan kernel work, headers: copy-block loop steps (pairs) %llu for %llu blocks | interval loop steps %llu for %llu intervals\n", acc[28], acc[30], acc[29], acc[31]);
#endif
    if (res) {
        res->arcs = acc[0]; res->chk = acc[1]; res->nodes = acc[2];
        res->kernel_ms = kernel_ms; res->launches = launches; res->slow_blocks = slow_blocks; res->lean_blocks = lean_blocks;
        res->index_bytes = (uint64_t)(to - from + 1) * (sh->offs.lo ? 4 : 8) + (sh->offs.lo ? ((uint64_t)(to - from) >> kOffShift) * 8 : 0) + (uint64_t)nblocks * 20;
        res->index_entries = a.skip_first && skx->h_first.size() > (size_t)lo + nblocks ? skx->h_first[lo + nblocks] - skx->h_first[lo] : 0;
        if (a.skip_first) res->index_bytes += res->index_entries * (2 + esz) + (uint64_t)nblocks * 9;
        res->graph_bytes = 0;
    }
    if (acc[3] && dbg_on()) fprintf(stderr, "[bvg] error bits 0x%llx\n", acc[3]);
    if (acc[3] & ERR_REF_RANGE) return BVG_E_STATE;
    if (acc[3] & (ERR_OVERRUN | ERR_MALFORMED)) return BVG_E_EOF;
    return 0;
}

// one entry of the index on the host
int read_offset(const Shared* sh, int64_t x, uint64_t* out) {
    if (!sh->offs.lo) { HIPCHK(hipMemcpy(out, sh->offs.wide + x, sizeof(uint64_t), hipMemcpyDeviceToHost)); return 0; }
    uint32_t lo = 0; uint64_t hi = 0;
    HIPCHK(hipMemcpy(&lo, sh->offs.lo + x, sizeof lo, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&hi, sh->offs.hi + (x >> kOffShift), sizeof hi, hipMemcpyDeviceToHost));
    *out = hi + lo;
    return 0;
}

// Packs the index (n+1 entries; on the device or on the host) into sh->offs.  1 = a distance does not fit 32 bits: the caller keeps
// the plain array.  A host array is staged through a 128 MiB device buffer, so the plain form never exists in HBM.
int pack_offsets(Shared* sh, const uint64_t* src_dev, const uint64_t* src_host) {
    const int64_t n1 = sh->p.nodes + 1, G = (int64_t)1 << kOffShift;
    DevBuf lo, hi, ovf, stagebuf;
    if (lo.alloc((size_t)n1 * sizeof(uint32_t)) || hi.alloc((size_t)((n1 + G - 1) / G + 1) * sizeof(uint64_t)) || ovf.alloc(sizeof(unsigned))) return BVG_E_NOMEM;
    HIPCHK(hipMemset(ovf.p, 0, sizeof(unsigned)));
    if (src_dev) launch_pack_offsets(src_dev, 0, n1, (uint32_t*)lo.p, (uint64_t*)hi.p, (unsigned*)ovf.p, nullptr);
    else {
        const int64_t step = (int64_t)1 << 24;
        if (stagebuf.alloc((size_t)std::min<int64_t>(step, n1) * sizeof(uint64_t))) return BVG_E_NOMEM;
        for (int64_t first = 0; first < n1; first += step) {
            const int64_t cnt = std::min<int64_t>(step, n1 - first);
            HIPCHK(hipMemcpy(stagebuf.p, src_host + first, (size_t)cnt * sizeof(uint64_t), hipMemcpyHostToDevice));
            launch_pack_offsets((const uint64_t*)stagebuf.p, first, cnt, (uint32_t*)lo.p, (uint64_t*)hi.p, (unsigned*)ovf.p, nullptr);
            HIPCHK(hipStreamSynchronize(nullptr));
        }
    }
    unsigned o = 0;
    HIPCHK(hipMemcpy(&o, ovf.p, sizeof o, hipMemcpyDeviceToHost));
    if (o) return 1;
    sh->d_off_lo = (uint32_t*)lo.release(); sh->d_off_hi = (uint64_t*)hi.release();
    sh->offs = Offsets{sh->d_off_lo, sh->d_off_hi, nullptr};
    return 0;
}

struct PackedOffsets { uint32_t* lo; uint64_t* hi; };   // bvg_tile hands over an index it wrote in packed form

int open_common(const bvg_params* p, const uint8_t* h_graph, const void* d_graph_in, uint64_t nbytes, const uint64_t* h_offsets,
                const void* d_offsets_in, int device, bvg_graph** out, const PackedOffsets* packed = nullptr) {
    if (!p || !out) return BVG_E_ARG;
    int r = check_params(*p); if (r) return r;
    r = ensure_device(device); if (r) return r;
    Shared* sh = new Shared();
    sh->device = device; sh->p = *p; sh->nbytes = nbytes;
    // 32-bit successor arithmetic holds every node id below 2^32 - 1 (0xFFFFFFFF is the lists' sentinel); the reference's own line between
    // the int and the long library is 2^31 because Java ints are signed -- nothing here is
    sh->wide = p->nodes > (int64_t)0xFFFFFF00ll || (knob("BVG_WIDE_FROM_2_31") != nullptr && p->nodes > (int64_t)0x7FFFFFFF);
    const int64_t n = p->nodes;
    if (d_graph_in) { sh->d_graph = (uint8_t*)d_graph_in; sh->own_graph = false; sh->padded = ((nbytes + 15) & ~15ull) + 16; }
    else {
        uint64_t padded = ((nbytes + 15) & ~15ull) + kPad;
        sh->padded = padded;
        HIPCHK(hipMalloc(&sh->d_graph, padded));
        sh->own_graph = true;
        HIPCHK(hipMemset(sh->d_graph, 0, padded));
        if (nbytes) HIPCHK(hipMemcpy(sh->d_graph, h_graph, nbytes, hipMemcpyHostToDevice));
    }
    // The index is kept packed (bvg_kernels.h: Offsets).  A caller's device array is packed into memory of our own and not referenced
    // afterwards; BVG_WIDE_OFFSETS=1 or a distance that does not fit 32 bits keeps the plain 64-bit form.
    const bool keep_wide = knob("BVG_WIDE_OFFSETS") != nullptr;
    if (packed) { sh->d_off_lo = packed->lo; sh->d_off_hi = packed->hi; sh->offs = Offsets{packed->lo, packed->hi, nullptr}; }
    else if (d_offsets_in) {
        int pk = keep_wide ? 1 : pack_offsets(sh, (const uint64_t*)d_offsets_in, nullptr);
        if (pk < 0) { release_shared(sh); return pk; }
        if (pk) { sh->d_off_wide = (uint64_t*)d_offsets_in; sh->own_wide = false; sh->offs = Offsets{nullptr, nullptr, sh->d_off_wide}; }
    } else if (h_offsets) {
        int pk = keep_wide ? 1 : pack_offsets(sh, nullptr, h_offsets);
        if (pk < 0) { release_shared(sh); return pk; }
        if (pk) {
            HIPCHK(hipMalloc(&sh->d_off_wide, ((size_t)n + 1) * sizeof(uint64_t)));
            sh->own_wide = true; sh->offs = Offsets{nullptr, nullptr, sh->d_off_wide};
            HIPCHK(hipMemcpy(sh->d_off_wide, h_offsets, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        }
    } else {
        uint64_t* d_wide = nullptr;
        HIPCHK(hipMalloc(&d_wide, ((size_t)n + 1) * sizeof(uint64_t)));
        sh->d_off_wide = d_wide; sh->own_wide = true; sh->offs = Offsets{nullptr, nullptr, d_wide};
        {
            // no .offsets (loadSequential / loadOffline, BVG:1345-1464; BVGraph -O, BVG:2595-2609): derive the index from
            // the stream itself with one sequential pass on the device
            unsigned* d_err = nullptr;
            HIPCHK(hipMalloc(&d_err, sizeof(unsigned)));
            HIPCHK(hipMemset(d_err, 0, sizeof(unsigned)));
            // Default: the chunk-parallel walk of bvg_derive.hip (round 3: one code per lane and step, only changed chunks re-walked).
            // Fall-back -- windows > 127, any oddity in the stream, BVG_DERIVE_SEQ=1 -- is the one-wavefront sequential walk, whose error
            // bits are the documented ones.
            int rounds = 0;
            int pr = knob("BVG_DERIVE_SEQ") ? -1 : derive_offsets_parallel(sh->d_graph, nbytes, n, p->window_size, p->min_interval_length, codings_of(*p), d_wide, d_err, nullptr, &rounds);
            if (pr == 0) {
                unsigned e0 = 0;
                if (hipMemcpy(&e0, d_err, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(d_err); release_shared(sh); return BVG_E_HIP; }
                if (e0) { pr = -4; (void)hipMemset(d_err, 0, sizeof(unsigned)); }
            }
            if (dbg_on()) fprintf(stderr, "[bvg] derive offsets: parallel walk %s (%d rounds)\n", pr == 0 ? "ok" : "not used / failed", rounds);
            if (pr != 0) launch_derive_offsets(sh->d_graph, sh->padded, nbytes, n, p->window_size, p->min_interval_length, codings_of(*p), d_wide, d_err, nullptr);
            unsigned herr = 0;
            hipError_t e = hipMemcpy(&herr, d_err, sizeof(unsigned), hipMemcpyDeviceToHost);
            (void)hipFree(d_err);
            if (e != hipSuccess) { release_shared(sh); return BVG_E_HIP; }
            if (dbg_on()) { uint64_t last = 0; (void)hipMemcpy(&last, d_wide + n, 8, hipMemcpyDeviceToHost); fprintf(stderr, "[bvg] derive offsets: err=%u end=%llu of %llu bits\n", herr, (unsigned long long)last, (unsigned long long)nbytes * 8); }
            if (herr) { release_shared(sh); return (herr & ERR_REF_RANGE) ? BVG_E_STATE : BVG_E_EOF; }
        }
        int pk = keep_wide ? 1 : pack_offsets(sh, d_wide, nullptr);
        if (pk < 0) { release_shared(sh); return pk; }
        if (pk == 0) { (void)hipFree(d_wide); sh->d_off_wide = nullptr; sh->own_wide = false; }
    }
    r = read_offset(sh, n, &sh->total_bits); if (r) { release_shared(sh); return r; }
    if (sh->total_bits > nbytes * 8) { release_shared(sh); return BVG_E_EOF; }
    r = make_handle(sh, out);
    if (r) { release_shared(sh); return r; }
    return 0;
}

}  // namespace

// ================================================================================================
// No C++ exception may cross the C ABI (a JVM behind JNI would be torn down by std::terminate): entry points that allocate
// host memory run inside this guard.
template <typename F> static int guarded(F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return BVG_E_NOMEM; }
    catch (const std::length_error&) { return BVG_E_ARG; }
    catch (...) { return BVG_E_STATE; }
}

// ---- the device index on disk (basename.bvgidx) ----
// What a first scan builds -- the block plan (boundaries, halos, largest lists) and the residual skip index with its validation marks
// -- written next to the graph so that the next process loads it instead of scanning the graph twice (the reference caches its own
// index the same way: the .obl file of the offsets big list, checked against the file it was built from, BVG:1545-1555).
// The lean scan kernel trusts the marks (it skips the checks a validated block cannot fail), so the file is tied to the graph by MORE
// than size and date (format version 2): a hash of EVERY byte of the stream (one pass on the device), every parameter that shapes a
// record (window, minimum interval length, zeta k, the five codings), and a checksum over the whole payload; every array is range-checked
// on the way in (halo lengths, marks, monotone entry counts, sizes).  A file that fails any of it is refused (BVG_E_IO) and the index is
// built from the stream as usual.
struct IndexHeader {
    char magic[8]; uint32_t version, block_bits; uint64_t graph_bytes, total_bits; int64_t nodes; uint64_t stream_hash;
    uint32_t window, wide, nblk, has_skip, skip_lo, skip_hi; uint64_t skip_total;
    int32_t min_interval, zeta_k, cod_outdegree, cod_block, cod_residual, cod_reference, cod_block_count; uint32_t skip_min, skip_every, pad0;
    uint64_t payload_hash;                      // of everything behind the header, array by array (host arrays on the host, device arrays on the device)
};
static const char kIndexMagic[8] = {'B', 'V', 'G', 'I', 'D', 'X', '2', 0};

// position-keyed word hash of a device array (launch_hash_words), synchronous
static int device_hash(bvg_graph* g, const void* d, uint64_t bytes, uint64_t* out) {
    DevBuf acc;
    if (acc.alloc(8)) return BVG_E_NOMEM;
    HIPCHK(hipMemsetAsync(acc.p, 0, 8, g->stream));
    if (bytes) launch_hash_words(d, bytes, (unsigned long long*)acc.p, g->stream);
    HIPCHK(hipMemcpyAsync(out, acc.p, 8, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return 0;
}
// the same function on the host (the small per-block arrays never leave it)
static uint64_t host_hash(const void* p, uint64_t nbytes) {
    const uint8_t* b = (const uint8_t*)p; const uint64_t nw = nbytes >> 3; uint64_t h = 0;
    for (uint64_t i = 0; i < nw; i++) { uint64_t w; memcpy(&w, b + 8 * i, 8); uint64_t z = w + (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 27; h += z; }
    if (nbytes & 7) { uint64_t t = 0; for (uint64_t k = nw << 3; k < nbytes; k++) t = (t << 8) | b[k]; uint64_t z = t + (nw + 1) * 0x9E3779B97F4A7C15ull + (nbytes & 7); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 27; h += z; }
    return h;
}
static inline uint64_t fold_hash(uint64_t acc, uint64_t part, uint64_t slot) { return (acc ^ (part + slot * 0xD6E8FEB86659FD93ull)) * 0xFF51AFD7ED558CCDull; }   // the arrays in order

static void fill_header_params(const Shared* sh, IndexHeader& h) {
    h.graph_bytes = sh->nbytes; h.total_bits = sh->total_bits; h.nodes = sh->p.nodes; h.window = (uint32_t)sh->p.window_size;
    h.min_interval = sh->p.min_interval_length; h.zeta_k = sh->p.zeta_k; h.cod_outdegree = sh->p.outdegree_coding; h.cod_block = sh->p.block_coding;
    h.cod_residual = sh->p.residual_coding; h.cod_reference = sh->p.reference_coding; h.cod_block_count = sh->p.block_count_coding;
    h.skip_min = kSkipMin; h.skip_every = kSkipEvery;          // (the index's own granularity when the file holds one: save / load)
}
static bool put_dev(FILE* f, const void* d, size_t bytes) {               // device array -> file, in pieces
    std::vector<uint8_t> buf(std::min<size_t>(bytes ? bytes : 1, (size_t)64 << 20));
    for (size_t o = 0; o < bytes; o += buf.size()) {
        const size_t k = std::min(buf.size(), bytes - o);
        if (hipMemcpy(buf.data(), (const char*)d + o, k, hipMemcpyDeviceToHost) != hipSuccess || fwrite(buf.data(), 1, k, f) != k) return false;
    }
    return true;
}
static bool get_dev(FILE* f, void* d, size_t bytes) {                     // file -> device array
    std::vector<uint8_t> buf(std::min<size_t>(bytes ? bytes : 1, (size_t)64 << 20));
    for (size_t o = 0; o < bytes; o += buf.size()) {
        const size_t k = std::min(buf.size(), bytes - o);
        if (fread(buf.data(), 1, k, f) != k || hipMemcpy((char*)d + o, buf.data(), k, hipMemcpyHostToDevice) != hipSuccess) return false;
    }
    return true;
}

static int save_index_impl(bvg_graph* g, const char* path) {
    if (!g || !path) return BVG_E_ARG;
    Shared* sh = g->sh;
    HIPCHK(hipSetDevice(sh->device));
    std::shared_ptr<Plan> plp;
    int r = build_plan(g, block_bits_of(g), plp); if (r) return r;
    const Plan& pl = *plp;
    std::shared_ptr<SkipIndex> ix = std::atomic_load(&plp->skip);
    if (ix && ix->failed) ix.reset();                                      // (a failed build left nothing to save)
    IndexHeader h{};
    memcpy(h.magic, kIndexMagic, 8); h.version = 2; h.block_bits = pl.block_bits; fill_header_params(sh, h);
    r = device_hash(g, sh->d_graph, sh->nbytes, &h.stream_hash); if (r) return r;
    h.nblk = pl.nblk;
    const size_t nb = pl.nblk;
    if (pl.h_maxd.size() != nb) return BVG_E_STATE;
    if (ix) { h.has_skip = 1; h.wide = ix->wide ? 1u : 0u; h.skip_lo = ix->blk_lo; h.skip_hi = ix->blk_hi; h.skip_total = ix->total; h.skip_min = ix->skip_min; h.skip_every = 1u << ix->skip_shift; }
    {   // payload checksum: the arrays in file order
        uint64_t acc = 0, part = 0;
        acc = fold_hash(acc, host_hash(pl.h_first.data(), (nb + 1) * 8), 1); acc = fold_hash(acc, host_hash(pl.h_maxd.data(), nb * 4), 2);
        r = device_hash(g, pl.d_halo, nb * 4, &part); if (r) return r; acc = fold_hash(acc, part, 3);
        r = device_hash(g, pl.d_mask, nb * 8, &part); if (r) return r; acc = fold_hash(acc, part, 4);
        if (ix) {
            acc = fold_hash(acc, host_hash(ix->h_first.data(), (nb + 1) * 8), 5); acc = fold_hash(acc, host_hash(ix->h_fmt.data(), nb), 6);
            r = device_hash(g, ix->d_bit, ix->total * 2, &part); if (r) return r; acc = fold_hash(acc, part, 7);
            r = device_hash(g, ix->d_val, ix->total * (ix->wide ? 8 : 4), &part); if (r) return r; acc = fold_hash(acc, part, 8);
        }
        h.payload_hash = acc;
    }
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return BVG_E_IO;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && fwrite(pl.h_first.data(), 8, nb + 1, f) == nb + 1 && fwrite(pl.h_maxd.data(), 4, nb, f) == nb;
    ok = ok && put_dev(f, pl.d_halo, nb * 4) && put_dev(f, pl.d_mask, nb * 8);
    if (ok && ix) {
        ok = fwrite(ix->h_first.data(), 8, nb + 1, f) == nb + 1 && fwrite(ix->h_fmt.data(), 1, nb, f) == nb;
        ok = ok && put_dev(f, ix->d_bit, (size_t)ix->total * 2) && put_dev(f, ix->d_val, (size_t)ix->total * (ix->wide ? 8 : 4));
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return BVG_E_IO; }
    return 0;
}

// BVG_E_IO: no such file / not an index of this graph / damaged (the caller then simply builds the index as usual)
static int load_index_impl(bvg_graph* g, const char* path) {
    if (!g || !path) return BVG_E_ARG;
    Shared* sh = g->sh;
    HIPCHK(hipSetDevice(sh->device));
    FILE* f = fopen(path, "rb");
    if (!f) return BVG_E_IO;
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    IndexHeader h{}, want{};
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kIndexMagic, 8) != 0 || h.version != 2) return BVG_E_IO;
    fill_header_params(sh, want);
    if (h.graph_bytes != want.graph_bytes || h.total_bits != want.total_bits || h.nodes != want.nodes || h.window != want.window || h.min_interval != want.min_interval ||
        h.zeta_k != want.zeta_k || h.cod_outdegree != want.cod_outdegree || h.cod_block != want.cod_block || h.cod_residual != want.cod_residual ||
        h.cod_reference != want.cod_reference || h.cod_block_count != want.cod_block_count ||
        h.skip_min < 2u || h.skip_min > 4096u || h.skip_every < 2u || h.skip_every > 64u || (h.skip_every & (h.skip_every - 1u)) != 0 ||      // (the granularity is the file's own: any valid one)
        h.block_bits != block_bits_of(g) || h.nblk == 0 || (uint64_t)h.nblk > (uint64_t)sh->p.nodes || h.wide > 1u || h.has_skip > 1u) return BVG_E_IO;
    // sizes first: the file must hold exactly what the header promises (and the entry count must be one the stream could produce)
    const size_t nb = h.nblk;
    if (h.has_skip && (h.skip_lo >= h.skip_hi || h.skip_hi > h.nblk || h.skip_total > sh->total_bits)) return BVG_E_IO;
    {
        const uint64_t vb = h.wide ? 8 : 4;
        uint64_t want_bytes = sizeof h + (uint64_t)(nb + 1) * 8 + (uint64_t)nb * 4 + (uint64_t)nb * 4 + (uint64_t)nb * 8;
        if (h.has_skip) want_bytes += (uint64_t)(nb + 1) * 8 + nb + h.skip_total * 2 + h.skip_total * vb;
        if (fseek(f, 0, SEEK_END) != 0) return BVG_E_IO;
        const long long fsz = ftell(f);
        if (fsz < 0 || (uint64_t)fsz != want_bytes || fseek(f, (long)sizeof h, SEEK_SET) != 0) return BVG_E_IO;
    }
    {   // every byte of the stream, hashed on the device: a .graph rewritten in place with the same size is not this index's graph
        uint64_t sh_hash = 0;
        int r = device_hash(g, sh->d_graph, sh->nbytes, &sh_hash); if (r) return r;
        if (sh_hash != h.stream_hash) return BVG_E_IO;
    }
    std::shared_ptr<Plan> np = std::make_shared<Plan>();
    Plan& pl = *np;
    pl.device = sh->device; pl.block_bits = h.block_bits; pl.nblk = h.nblk; pl.h_first.resize(nb + 1); pl.h_maxd.resize(nb);
    if (fread(pl.h_first.data(), 8, nb + 1, f) != nb + 1 || fread(pl.h_maxd.data(), 4, nb, f) != nb) return BVG_E_IO;
    if (pl.h_first[0] != 0 || pl.h_first[nb] != (uint64_t)sh->p.nodes) return BVG_E_IO;
    for (size_t i = 0; i < nb; i++) if (pl.h_first[i] >= pl.h_first[i + 1]) return BVG_E_IO;
    uint64_t acc = 0, part = 0;
    acc = fold_hash(acc, host_hash(pl.h_first.data(), (nb + 1) * 8), 1); acc = fold_hash(acc, host_hash(pl.h_maxd.data(), nb * 4), 2);
    if (hipMalloc(&pl.d_first, (nb + 1) * 8) != hipSuccess || hipMalloc(&pl.d_halo, nb * 4) != hipSuccess || hipMalloc(&pl.d_mask, nb * 8) != hipSuccess) { (void)hipGetLastError(); return BVG_E_NOMEM; }
    HIPCHK(hipMemcpy(pl.d_first, pl.h_first.data(), (nb + 1) * 8, hipMemcpyHostToDevice));
    {   // halos: range-checked on the host on their way in (a halo reaches at most kMaxHalo nodes back and never before node 0)
        std::vector<uint32_t> halo(nb);
        if (fread(halo.data(), 4, nb, f) != nb) return BVG_E_IO;
        for (size_t i = 0; i < nb; i++) if (halo[i] > (uint32_t)kMaxHalo || (uint64_t)halo[i] > pl.h_first[i]) return BVG_E_IO;
        HIPCHK(hipMemcpy(pl.d_halo, halo.data(), nb * 4, hipMemcpyHostToDevice));
        acc = fold_hash(acc, host_hash(halo.data(), nb * 4), 3);
    }
    if (!get_dev(f, pl.d_mask, nb * 8)) return BVG_E_IO;
    { int r = device_hash(g, pl.d_mask, nb * 8, &part); if (r) return r; acc = fold_hash(acc, part, 4); }
    std::shared_ptr<SkipIndex> ix;
    if (h.has_skip) {
        ix = std::make_shared<SkipIndex>();
        ix->device = sh->device; ix->blk_lo = h.skip_lo; ix->blk_hi = h.skip_hi; ix->total = h.skip_total; ix->wide = h.wide != 0; ix->gen = next_plan_version();
        ix->skip_min = h.skip_min; ix->skip_shift = 0; while ((1u << ix->skip_shift) < h.skip_every) ix->skip_shift++;
        ix->h_first.resize(nb + 1); ix->h_fmt.resize(nb);
        if (fread(ix->h_first.data(), 8, nb + 1, f) != nb + 1 || fread(ix->h_fmt.data(), 1, nb, f) != nb) return BVG_E_IO;
        if (ix->h_first[0] != 0 || ix->h_first[nb] != ix->total) return BVG_E_IO;
        for (size_t i = 0; i < nb; i++) {
            if (ix->h_first[i] > ix->h_first[i + 1] || ix->h_fmt[i] > 3) return BVG_E_IO;
            if ((i < ix->blk_lo || i >= ix->blk_hi) && (ix->h_first[i] != ix->h_first[i + 1] || ix->h_fmt[i] != 0)) return BVG_E_IO;   // nothing outside the indexed blocks
        }
        acc = fold_hash(acc, host_hash(ix->h_first.data(), (nb + 1) * 8), 5); acc = fold_hash(acc, host_hash(ix->h_fmt.data(), nb), 6);
        const size_t vb = ix->wide ? 8 : 4;
        if (hipMalloc(&ix->d_first, (nb + 1) * 8) != hipSuccess || hipMalloc(&ix->d_bit, (size_t)ix->total * 2 + 16) != hipSuccess || hipMalloc(&ix->d_fmt, nb) != hipSuccess ||
            hipMalloc(&ix->d_val, (size_t)ix->total * vb + 16) != hipSuccess) { (void)hipGetLastError(); return BVG_E_NOMEM; }
        HIPCHK(hipMemcpy(ix->d_first, ix->h_first.data(), (nb + 1) * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ix->d_fmt, ix->h_fmt.data(), nb, hipMemcpyHostToDevice));
        if (!get_dev(f, ix->d_bit, (size_t)ix->total * 2) || !get_dev(f, ix->d_val, (size_t)ix->total * vb)) return BVG_E_IO;
        int r = device_hash(g, ix->d_bit, ix->total * 2, &part); if (r) return r; acc = fold_hash(acc, part, 7);
        r = device_hash(g, ix->d_val, ix->total * vb, &part); if (r) return r; acc = fold_hash(acc, part, 8);
    }
    if (acc != h.payload_hash) return BVG_E_IO;                              // bit rot, truncation that kept the size, an edited file
    if (ix) std::atomic_store(&pl.skip, ix);
    pl.version = next_plan_version();
    std::lock_guard<std::mutex> lk(sh->mu);
    sh->plans.clear(); sh->plans[pl.block_bits] = np;
    if (dbg_on()) fprintf(stderr, "[bvg] index loaded from %s: %u blocks, %llu skip entries\n", path, h.nblk, (unsigned long long)h.skip_total);
    return 0;
}

extern "C" {

int bvg_abi_version(void) { return BVG_ABI_VERSION; }

void bvg_default_params(bvg_params* p) {
    memset(p, 0, sizeof *p);
    p->arcs = -1;
    p->window_size = 7; p->max_ref_count = 3; p->min_interval_length = 4; p->zeta_k = 3;
    p->outdegree_coding = BVG_GAMMA; p->block_coding = BVG_GAMMA; p->residual_coding = BVG_ZETA;
    p->reference_coding = BVG_UNARY; p->block_count_coding = BVG_GAMMA; p->offset_coding = BVG_GAMMA;
}

int bvg_parse_properties(const char* text, size_t len, bvg_params* out) {
    if (!text || !out) return BVG_E_ARG;
    bvg_params p; bvg_default_params(&p);
    bool have_nodes = false, have_class = false; long version = 0;
    std::string t(text, len);
    size_t i = 0;
    auto trim = [](std::string s) {
        size_t a = 0, b = s.size();
        while (a < b && isspace((unsigned char)s[a])) a++;
        while (b > a && isspace((unsigned char)s[b - 1])) b--;
        return s.substr(a, b - a);
    };
    while (i < t.size()) {
        size_t e = t.find_first_of("\r\n", i); if (e == std::string::npos) e = t.size();
        std::string line = trim(t.substr(i, e - i));
        i = e + 1;
        if (line.empty() || line[0] == '#' || line[0] == '!') continue;
        size_t sep = line.find_first_of("=:");
        std::string key = trim(sep == std::string::npos ? line : line.substr(0, sep));
        std::string val = sep == std::string::npos ? "" : trim(line.substr(sep + 1));
        if (key == "nodes") { p.nodes = strtoll(val.c_str(), nullptr, 10); have_nodes = true; }
        else if (key == "arcs") p.arcs = strtoll(val.c_str(), nullptr, 10);
        else if (key == "windowsize") p.window_size = (int32_t)strtol(val.c_str(), nullptr, 10);
        else if (key == "maxrefcount") p.max_ref_count = (int32_t)strtol(val.c_str(), nullptr, 10);
        else if (key == "minintervallength") p.min_interval_length = (int32_t)strtol(val.c_str(), nullptr, 10);
        else if (key == "zetak") p.zeta_k = (int32_t)strtol(val.c_str(), nullptr, 10);
        else if (key == "version") version = strtol(val.c_str(), nullptr, 10);
        else if (key == "graphclass") {
            if (val.rfind("class ", 0) == 0) val = val.substr(6);
            if (val != "it.unimi.dsi.big.webgraph.BVGraph" && val != "it.unimi.dsi.webgraph.BVGraph") return BVG_E_IO;   // BVG:1491
            have_class = true;
        } else if (key == "compressionflags") {
            size_t s = 0;
            while (s <= val.size()) {
                size_t b = val.find('|', s); if (b == std::string::npos) b = val.size();
                std::string f = trim(val.substr(s, b - s));
                s = b + 1;
                if (f.empty()) continue;
                static const struct { const char* prefix; int field; unsigned allowed; } F[] = {
                    {"OUTDEGREES_", 0, 1u << BVG_GAMMA | 1u << BVG_DELTA},
                    {"BLOCKS_", 1, 1u << BVG_GAMMA | 1u << BVG_DELTA},
                    {"RESIDUALS_", 2, 1u << BVG_GAMMA | 1u << BVG_ZETA | 1u << BVG_DELTA | 1u << BVG_NIBBLE | 1u << BVG_GOLOMB},
                    {"REFERENCES_", 3, 1u << BVG_GAMMA | 1u << BVG_DELTA | 1u << BVG_UNARY},
                    {"BLOCK_COUNT_", 4, 1u << BVG_GAMMA | 1u << BVG_DELTA | 1u << BVG_UNARY},
                    {"OFFSETS_", 5, 1u << BVG_GAMMA | 1u << BVG_DELTA}};
                static const struct { const char* name; int id; } N[] = {{"DELTA", BVG_DELTA}, {"GAMMA", BVG_GAMMA}, {"GOLOMB", BVG_GOLOMB},
                    {"SKEWED_GOLOMB", BVG_SKEWED_GOLOMB}, {"UNARY", BVG_UNARY}, {"ZETA", BVG_ZETA}, {"NIBBLE", BVG_NIBBLE}};
                bool ok = false;
                for (auto& fd : F) {
                    size_t pl = strlen(fd.prefix);
                    if (f.compare(0, pl, fd.prefix) != 0) continue;
                    std::string nm = f.substr(pl);
                    for (auto& nn : N) if (nm == nn.name && (fd.allowed >> nn.id & 1u)) {
                        int32_t* dst[] = {&p.outdegree_coding, &p.block_coding, &p.residual_coding, &p.reference_coding, &p.block_count_coding, &p.offset_coding};
                        *dst[fd.field] = nn.id; ok = true;
                    }
                    if (ok) break;
                }
                if (!ok) return BVG_E_IO;                                           // "Compression flag unknown", BVG:1326
            }
        }
    }
    if (!have_nodes || !have_class) return BVG_E_IO;
    if (version > 0) return BVG_E_IO;                                               // BVG:1496-1497
    *out = p;
    return 0;
}

int bvg_decode_offsets(const uint8_t* obytes, size_t nbytes, int64_t nodes, int coding, uint64_t* out) {
    if (!obytes || !out || nodes < 0) return BVG_E_ARG;
    if (coding != BVG_GAMMA && coding != BVG_DELTA) return BVG_E_UNSUPPORTED;      // BVG:628-632
    HostBits b{obytes, (uint64_t)nbytes * 8};
    uint64_t off = 0;
    for (int64_t i = 0; i <= nodes; i++) {                                          // n+1 gaps, BVG:885
        off += coding == BVG_DELTA ? b.delta() : b.gamma();
        if (b.eof) return BVG_E_EOF;
        out[i] = off;
    }
    return 0;
}

int bvg_open(const char* basename, int load_mode, int device, bvg_graph** out) {
    if (!basename || !out) return BVG_E_ARG;
    if (load_mode < BVG_LOAD_OFFLINE || load_mode > BVG_LOAD_MAPPED) return BVG_E_ARG;
    return guarded([&]() -> int {
    std::string base(basename);
    std::vector<uint8_t> props, graph, offs;
    int r = read_file(base + ".properties", props); if (r) return r;
    bvg_params p;
    r = bvg_parse_properties((const char*)props.data(), props.size(), &p); if (r) return r;
    r = check_params(p); if (r) return r;                        // before anything is sized from the file's own numbers
    r = read_file(base + ".graph", graph); if (r) return r;
    if ((uint64_t)p.nodes > (uint64_t)graph.size() * 8 + 1) return BVG_E_IO;   // every record takes at least one bit: a corrupt `nodes`
    // Standard / mapped loads read basename.offsets (BVG:1545-1558).  Sequential / offline loads (BVG:1345-1464) do not
    // have to have it: the index is then derived from the stream on the device.
    r = read_file(base + ".offsets", offs);
    if (r) {
        if (load_mode >= BVG_LOAD_STANDARD) return BVG_E_IO;
        return open_common(&p, graph.data(), nullptr, graph.size(), nullptr, nullptr, device, out);
    }
    std::vector<uint64_t> offsets((size_t)p.nodes + 1);
    r = bvg_decode_offsets(offs.data(), offs.size(), p.nodes, p.offset_coding, offsets.data()); if (r) return r;
    r = open_common(&p, graph.data(), nullptr, graph.size(), offsets.data(), nullptr, device, out);
    if (r == 0) {
        // a saved device index (bvg_save_index) that is not older than the graph is loaded instead of rebuilt (cf. the .obl cache, BVG:1545-1555);
        // anything wrong with it just means the index is built as usual
        struct stat sg {}, si {};
        const std::string ip = base + ".bvgidx";
        if (stat(ip.c_str(), &si) == 0 && stat((base + ".graph").c_str(), &sg) == 0 && si.st_mtime >= sg.st_mtime) (void)load_index_impl(*out, ip.c_str());
    }
    return r;
    });
}

int bvg_open_mem(const bvg_params* p, const uint8_t* graph, uint64_t nbytes, const uint64_t* offsets, int device, bvg_graph** out) {
    if (!graph && nbytes) return BVG_E_ARG;
    return open_common(p, graph, nullptr, nbytes, offsets, nullptr, device, out);
}

int bvg_open_dev(const bvg_params* p, const void* d_graph, uint64_t nbytes, const void* d_offsets, int device, bvg_graph** out) {
    if (!d_graph || !d_offsets) return BVG_E_ARG;
    return open_common(p, nullptr, d_graph, nbytes, nullptr, d_offsets, device, out);
}

int bvg_copy(const bvg_graph* g, bvg_graph** out) {
    if (!g || !out) return BVG_E_ARG;
    g->sh->refs.fetch_add(1);
    int r = make_handle(g->sh, out);
    if (r) { release_shared(g->sh); return r; }
    (*out)->node_base = g->node_base; (*out)->tun = g->tun;
    return 0;
}

void bvg_close(bvg_graph* g) {
    if (!g) return;
    (void)hipSetDevice(g->sh->device);
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->tr_ws) (void)hipFree(g->tr_ws);
    if (g->dr_ws) (void)hipFree(g->dr_ws);
    if (g->flow_ws) (void)hipFree(g->flow_ws);
    if (g->d_acc) (void)hipFree(g->d_acc);
    if (g->d_fail) (void)hipFree(g->d_fail);
    if (g->slow_ws) (void)hipFree(g->slow_ws);
    if (g->giant_ws) (void)hipFree(g->giant_ws);
    if (g->d_gslots) (void)hipFree(g->d_gslots);
    if (g->pred.d_lists) (void)hipFree(g->pred.d_lists);
    for (int i = 0; i < bvg_graph::kSide; i++) { if (g->side[i]) { (void)hipStreamSynchronize(g->side[i]); (void)hipStreamDestroy(g->side[i]); } if (g->side_ev[i]) (void)hipEventDestroy(g->side_ev[i]); }
    release_shared(g->sh);
    delete g;
}

int bvg_info(const bvg_graph* g, bvg_params* out) { if (!g || !out) return BVG_E_ARG; *out = g->sh->p; return 0; }
int bvg_set_node_base(bvg_graph* g, uint64_t node_base) { if (!g) return BVG_E_ARG; g->node_base = node_base; return 0; }
int bvg_set_tuning(bvg_graph* g, const bvg_tuning* t) { if (!g || !t) return BVG_E_ARG; g->tun = *t; return 0; }

int bvg_get_offsets(bvg_graph* g, uint64_t* out) {
    if (!g || !out) return BVG_E_ARG;
    HIPCHK(hipSetDevice(g->sh->device));
    const Shared* sh = g->sh; const int64_t n1 = sh->p.nodes + 1;
    if (!sh->offs.lo) { HIPCHK(hipMemcpy(out, sh->offs.wide, (size_t)n1 * sizeof(uint64_t), hipMemcpyDeviceToHost)); return 0; }
    DevBuf tmp;                                                        // unpacked in pieces through a 128 MiB device buffer
    const int64_t step = (int64_t)1 << 24;
    if (tmp.alloc((size_t)std::min<int64_t>(step, n1) * sizeof(uint64_t))) return BVG_E_NOMEM;
    for (int64_t first = 0; first < n1; first += step) {
        const int64_t cnt = std::min<int64_t>(step, n1 - first);
        launch_unpack_offsets(sh->offs, first, cnt, (uint64_t*)tmp.p, g->stream);
        HIPCHK(hipStreamSynchronize(g->stream));
        HIPCHK(hipMemcpy(out + first, tmp.p, (size_t)cnt * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int bvg_outdegrees(bvg_graph* g, int64_t from, int64_t to, int32_t* out) {
    if (!g || !out) return BVG_E_ARG;
    if (from < 0 || to > g->sh->p.nodes || from > to) return BVG_E_ARG;            // BVG:823
    if (from == to) return 0;
    HIPCHK(hipSetDevice(g->sh->device));
    int32_t* d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)(to - from) * sizeof(int32_t)));
    launch_outdegrees(g->sh->d_graph, g->sh->nbytes, g->sh->offs, from, to, g->sh->p.outdegree_coding, d, nullptr, g->stream);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)(to - from) * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    (void)hipFree(d);
    return e == hipSuccess ? 0 : BVG_E_HIP;
}

// Per-handle device workspace for the materialising calls: grown on demand, kept between calls (a NodeIterator asks for batch
// after batch of the same size; a fresh hipMalloc / hipFree pair per buffer and call cost more than the decode of a small batch).
static int dr_ensure(bvg_graph* g, size_t bytes) {
    if (bytes <= g->dr_ws_bytes) return 0;
    if (g->dr_ws) { (void)hipFree(g->dr_ws); g->dr_ws = nullptr; g->dr_ws_bytes = 0; }
    const size_t want = bytes + bytes / 4;
    if (hipMalloc(&g->dr_ws, want) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&g->dr_ws, bytes) != hipSuccess) { (void)hipGetLastError(); g->dr_ws = nullptr; return BVG_E_NOMEM; }
        g->dr_ws_bytes = bytes; return 0;
    }
    g->dr_ws_bytes = want;
    return 0;
}

static int decode_range_impl(bvg_graph* g, int64_t from, int64_t to, int32_t* outdeg, int64_t* succ, uint64_t cap, uint64_t* n_succ, bool dev, bool narrow = false) {
    if (!g) return BVG_E_ARG;
    Shared* sh = g->sh;
    if (narrow && (dev || (uint64_t)sh->p.nodes + g->node_base > 0xFFFFFFFFull)) return BVG_E_UNSUPPORTED;   // 32-bit ids: host path, every id below 2^32 - 1 (0xFFFFFFFF stands for the -1 of a malformed stream, never for a node)
    if (from < 0 || to > sh->p.nodes || from > to) return BVG_E_ARG;               // BVG:863,1000,1128
    if (from == to) { if (n_succ) *n_succ = 0; return 0; }
    HIPCHK(hipSetDevice(sh->device));
    const int64_t cnt = to - from;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // workspace: [cum | scan tmp | deg (unless the caller's device buffer takes them)] and, for host callers, the successors behind them
    const size_t o_cum = 0, o_tmp = o_cum + al(((size_t)cnt + 1) * sizeof(uint64_t)), o_deg = o_tmp + al(scan_tmp_elems(cnt) * sizeof(uint64_t));
    const size_t o_succ = o_deg + al((size_t)cnt * sizeof(int32_t));
    int rc = dr_ensure(g, o_succ); if (rc) return rc;
    auto at = [&](size_t off) { return (char*)g->dr_ws + off; };
    int32_t* d_deg = (dev && outdeg) ? outdeg : (int32_t*)at(o_deg);
    uint64_t* d_cum = (uint64_t*)at(o_cum);
    launch_outdegrees(sh->d_graph, sh->nbytes, sh->offs, from, to, sh->p.outdegree_coding, d_deg, nullptr, g->stream);
    launch_exclusive_scan(d_deg, d_cum, cnt, (uint64_t*)at(o_tmp), g->stream);
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_cum + cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    if (n_succ) *n_succ = total;
    if (total > cap || (!succ && total > 0)) {          // query / too small: report the size (and the outdegrees)
        if (outdeg && !dev) HIPCHK(hipMemcpy(outdeg, d_deg, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
        return BVG_E_CAPACITY;
    }
    int64_t* d_succ = succ;
    if (!dev) {
        const size_t per = sizeof(int64_t) + (narrow ? sizeof(uint32_t) : 0);
        if (o_succ + (size_t)(total ? total : 1) * per + 256 > g->dr_ws_bytes) {
            // growing moves the workspace: the prefix sums are recomputed rather than copied (two tiny kernels)
            rc = dr_ensure(g, o_succ + (size_t)(total ? total : 1) * per + 256); if (rc) return rc;
            d_deg = (int32_t*)at(o_deg); d_cum = (uint64_t*)at(o_cum);
            launch_outdegrees(sh->d_graph, sh->nbytes, sh->offs, from, to, sh->p.outdegree_coding, d_deg, nullptr, g->stream);
            launch_exclusive_scan(d_deg, d_cum, cnt, (uint64_t*)at(o_tmp), g->stream);
        }
        d_succ = (int64_t*)at(o_succ);
    }
    rc = run_decode(g, from, to, true, d_cum, d_succ, d_deg, nullptr);
    if (rc == 0 && !dev) {
        // device -> host on the handle's stream: at PCIe rate when the caller's buffers are page-locked (bvg_host_alloc)
        if (total && narrow) {
            uint32_t* d32 = (uint32_t*)at(o_succ + al((size_t)total * sizeof(int64_t)));
            launch_narrow_succ(d_succ, d32, total, g->stream);
            HIPCHK(hipMemcpyAsync(succ, d32, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, g->stream));
        } else if (total) HIPCHK(hipMemcpyAsync(succ, d_succ, (size_t)total * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
        if (outdeg) HIPCHK(hipMemcpyAsync(outdeg, d_deg, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
    }
    return rc;
}

int bvg_decode_range(bvg_graph* g, int64_t from, int64_t to, int32_t* outdeg, int64_t* succ, uint64_t succ_cap, uint64_t* n_succ) {
    return guarded([&] { return decode_range_impl(g, from, to, outdeg, succ, succ_cap, n_succ, false); });
}
int bvg_decode_range32(bvg_graph* g, int64_t from, int64_t to, int32_t* outdeg, uint32_t* succ, uint64_t succ_cap, uint64_t* n_succ) {
    return guarded([&] { return decode_range_impl(g, from, to, outdeg, (int64_t*)succ, succ_cap, n_succ, false, true); });
}
int bvg_decode_range_dev(bvg_graph* g, int64_t from, int64_t to, void* d_outdeg, void* d_succ, uint64_t succ_cap, uint64_t* n_succ) {
    return guarded([&] { return decode_range_impl(g, from, to, (int32_t*)d_outdeg, (int64_t*)d_succ, succ_cap, n_succ, true); });
}

void* bvg_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void bvg_host_free(void* p) { if (p) (void)hipHostFree(p); }

static int bvg_successors_batch_impl(bvg_graph* g, const int64_t* nodes, int64_t count, int32_t* outdeg, int64_t* succ, uint64_t succ_cap, uint64_t* n_succ) {
    if (!g || (!nodes && count) || count < 0) return BVG_E_ARG;
    Shared* sh = g->sh;
    for (int64_t i = 0; i < count; i++) if (nodes[i] < 0 || nodes[i] >= sh->p.nodes) return BVG_E_ARG;      // BVG:863
    if (n_succ) *n_succ = 0;
    if (count == 0) return 0;
    if (count > 0x3FFFFFFF) return BVG_E_ARG;
    HIPCHK(hipSetDevice(sh->device));
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t c = (size_t)count;
    const size_t o_nodes = 0, o_deg = o_nodes + al(c * sizeof(int64_t)), o_first = o_deg + al(c * sizeof(int32_t)), o_cum = o_first + al((2 * c + 1) * sizeof(uint64_t));
    const size_t o_tmp = o_cum + al((c + 1) * sizeof(uint64_t)), o_halo = o_tmp + al(scan_tmp_elems(count) * sizeof(uint64_t)), o_mask = o_halo + al(2 * c * sizeof(uint32_t));
    const size_t o_succ = o_mask + al(2 * c * sizeof(uint64_t));
    int rc = dr_ensure(g, o_succ); if (rc) return rc;
    auto at = [&](size_t off) { return (char*)g->dr_ws + off; };
    auto prepare = [&]() -> int {
        HIPCHK(hipMemcpyAsync(at(o_nodes), nodes, c * sizeof(int64_t), hipMemcpyHostToDevice, g->stream));
        launch_outdegrees_gather(sh->d_graph, sh->nbytes, sh->offs, (const int64_t*)at(o_nodes), count, sh->p.outdegree_coding, (int32_t*)at(o_deg), (uint64_t*)at(o_first), g->stream);
        launch_exclusive_scan((const int32_t*)at(o_deg), (uint64_t*)at(o_cum), count, (uint64_t*)at(o_tmp), g->stream);
        launch_plan_halo(sh->d_graph, sh->nbytes, sh->offs, sh->p.nodes, (const uint64_t*)at(o_first), (uint32_t)(2 * count), sh->p.window_size, codings_of(sh->p), (uint32_t*)at(o_halo), (uint64_t*)at(o_mask), g->stream);
        return 0;
    };
    rc = prepare(); if (rc) return rc;
    uint64_t total = 0;
    std::vector<uint32_t> halo(2 * c);
    HIPCHK(hipMemcpyAsync(&total, (uint64_t*)at(o_cum) + count, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipMemcpyAsync(halo.data(), at(o_halo), halo.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    if (n_succ) *n_succ = total;
    if (outdeg) HIPCHK(hipMemcpy(outdeg, at(o_deg), c * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (total > succ_cap || (!succ && total > 0)) return BVG_E_CAPACITY;
    // A request whose reference chain reaches more than 64 nodes back (possible with maxrefcount x window > 64) does not fit a
    // request block's halo: it is taken out of the batch (an empty block) and decoded afterwards through the graph's block plan,
    // whose blocks are cut so that every chain fits (successors(x) recurses as deep as the chain goes, BVG:1084).
    std::vector<int64_t> deep;
    for (int64_t i = 0; i < count; i++) if (halo[2 * (size_t)i] == 0xFFFFFFFFu) deep.push_back(i);
    std::vector<uint64_t> hcum;
    if (!deep.empty()) {
        hcum.resize(c + 1);
        HIPCHK(hipMemcpy(hcum.data(), at(o_cum), (c + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    auto blank_deep = [&]() -> int {
        for (int64_t i : deep) {
            const uint64_t pair[2] = {(uint64_t)nodes[i], (uint64_t)nodes[i]}; const uint32_t hz[2] = {0u, 0u};
            HIPCHK(hipMemcpy((uint64_t*)at(o_first) + 2 * (size_t)i, pair, sizeof pair, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy((uint32_t*)at(o_halo) + 2 * (size_t)i, hz, sizeof hz, hipMemcpyHostToDevice));
        }
        return 0;
    };
    rc = blank_deep(); if (rc) return rc;
    if (o_succ + (size_t)(total ? total : 1) * sizeof(int64_t) > g->dr_ws_bytes) {
        rc = dr_ensure(g, o_succ + (size_t)(total ? total : 1) * sizeof(int64_t)); if (rc) return rc;
        rc = prepare(); if (rc) return rc;                                   // the workspace moved: redo the (cheap) preparation in the new one
        rc = blank_deep(); if (rc) return rc;
    }
    BatchPlan bp{(const uint64_t*)at(o_first), (const uint32_t*)at(o_halo), (const uint64_t*)at(o_mask), (uint32_t)count};
    rc = run_decode(g, 0, sh->p.nodes, true, (const uint64_t*)at(o_cum), (int64_t*)at(o_succ), nullptr, nullptr, &bp);
    if (rc == 0 && total) {
        HIPCHK(hipMemcpyAsync(succ, at(o_succ), (size_t)total * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
    }
    for (size_t k = 0; k < deep.size() && rc == 0; k++) {                        // (the range decode reuses the workspace: the batch's results are on the host by now)
        const int64_t i = deep[k]; const uint64_t want = hcum[(size_t)i + 1] - hcum[(size_t)i];
        int32_t d1 = 0; uint64_t got = 0; int64_t dummy = 0;
        rc = decode_range_impl(g, nodes[i], nodes[i] + 1, &d1, want ? succ + hcum[(size_t)i] : &dummy, want ? want : 1, &got, false);
        if (rc == 0 && got != want) rc = BVG_E_STATE;
    }
    return rc;
}

static int bvg_scan_impl(bvg_graph* g, int64_t from, int64_t to, bvg_scan_result* out) {
    if (!g || !out) return BVG_E_ARG;
    Shared* sh = g->sh;
    if (from < 0 || to > sh->p.nodes || from > to) return BVG_E_ARG;
    memset(out, 0, sizeof *out);
    if (from == to) return 0;
    HIPCHK(hipSetDevice(sh->device));
    int r = run_decode(g, from, to, false, nullptr, nullptr, nullptr, out);
    // algorithmic bytes: the compressed bytes covering [from,to)
    uint64_t b[2];
    { int r2 = read_offset(sh, from, &b[0]); if (!r2) r2 = read_offset(sh, to, &b[1]); if (r2) return r2; }
    out->graph_bytes = (b[1] + 7) / 8 - b[0] / 8;
    return r;
}

static int transpose_impl(bvg_graph* g, uint64_t* toffsets, int64_t* tsucc, uint64_t cap, uint64_t* n_arcs, bool dev) {
    if (!g || !toffsets) return BVG_E_ARG;
    Shared* sh = g->sh;
    if (g->node_base != 0) return BVG_E_ARG;                 // a shard's targets leave its node range: transpose the whole graph
    const int64_t n = sh->p.nodes;
    HIPCHK(hipSetDevice(sh->device));
    const bool dbgt = dbg_on();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto tA = now();
    // One workspace per handle, grown on demand and kept between calls (a fresh multi-gigabyte hipMalloc costs far more
    // than the decode and the sort together): [deg | cum | scan tmp | bad] first, the arc-sized part once the arc count is known.
    const size_t nn = (size_t)(n > 0 ? n : 1);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_deg = 0, o_cum = o_deg + al(nn * sizeof(int32_t)), o_tmp = o_cum + al((nn + 1) * sizeof(uint64_t));
    const size_t o_bad = o_tmp + al(scan_tmp_elems((int64_t)nn) * sizeof(uint64_t)), o_arcs = o_bad + 256;
    auto ensure = [&](size_t bytes) -> int {
        if (bytes <= g->tr_ws_bytes) return 0;
        if (g->tr_ws) { (void)hipFree(g->tr_ws); g->tr_ws = nullptr; g->tr_ws_bytes = 0; }
        if (hipMalloc(&g->tr_ws, bytes) != hipSuccess) { (void)hipGetLastError(); return BVG_E_NOMEM; }
        g->tr_ws_bytes = bytes;
        return 0;
    };
    int rc = ensure(o_arcs); if (rc) return rc;
    auto at = [&](size_t off) { return (char*)g->tr_ws + off; };
    HIPCHK(hipMemsetAsync(at(o_bad), 0, sizeof(unsigned), g->stream));
    HIPCHK(hipMemsetAsync(at(o_cum), 0, (nn + 1) * sizeof(uint64_t), g->stream));
    uint64_t total = 0;
    if (n > 0) {
        launch_outdegrees(sh->d_graph, sh->nbytes, sh->offs, 0, n, sh->p.outdegree_coding, (int32_t*)at(o_deg), nullptr, g->stream);
        launch_exclusive_scan((const int32_t*)at(o_deg), (uint64_t*)at(o_cum), n, (uint64_t*)at(o_tmp), g->stream);
        HIPCHK(hipMemcpyAsync(&total, (uint64_t*)at(o_cum) + n, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    }
    HIPCHK(hipStreamSynchronize(g->stream));
    if (n_arcs) *n_arcs = total;
    if (total > cap || (!tsucc && total > 0)) return BVG_E_CAPACITY;
    const size_t mm = (size_t)(total ? total : 1);
    const size_t temp_b = transpose_temp_bytes(total, n > 0 ? n : 1);
    const size_t o_succ = o_arcs, o_src = o_succ + al(mm * 8), o_keys = o_src + al(mm * 8), o_temp = o_keys + al(mm * 8);
    const size_t o_toff = o_temp + al(temp_b ? temp_b : 16), o_ts = o_toff + (dev ? 0 : al((nn + 1) * 8)), o_end = o_ts + (dev ? 0 : al(mm * 8));
    {   // growing the workspace must not lose the prefix sums: save them on the host side of the copy only when it really grows
        if (o_end > g->tr_ws_bytes) {
            std::vector<char> keep(o_arcs);
            HIPCHK(hipMemcpy(keep.data(), g->tr_ws, o_arcs, hipMemcpyDeviceToHost));
            rc = ensure(o_end); if (rc) return rc;
            HIPCHK(hipMemcpy(g->tr_ws, keep.data(), o_arcs, hipMemcpyHostToDevice));
        }
    }
    uint64_t* const d_cum = (uint64_t*)at(o_cum); int64_t* const d_succ = (int64_t*)at(o_succ);
    g->tr_o_cum = o_cum; g->tr_o_succ = o_succ;
    uint64_t* const d_toff = dev ? toffsets : (uint64_t*)at(o_toff); int64_t* const d_tsucc = dev ? tsucc : (int64_t*)at(o_ts);
    const auto tB = now();
    if (n > 0) {                                             // the decode: every successor list, source-major, stays in HBM
        rc = run_decode(g, 0, n, true, d_cum, d_succ, nullptr, nullptr);
        if (rc) return rc;
    }
    const auto tC = now();
    if (transpose_pairs(d_cum, n, total, d_succ, (int64_t*)at(o_src), (uint64_t*)at(o_keys), at(o_temp), temp_b, d_toff, d_tsucc, (unsigned*)at(o_bad), g->stream) != hipSuccess) {
        (void)hipGetLastError(); return BVG_E_HIP;
    }
    unsigned bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, at(o_bad), sizeof(unsigned), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    const auto tD = now();
    if (dbgt) fprintf(stderr, "[bvg] transpose: outdegrees + workspace %.1f ms, decode %.1f ms, expand + sort + offsets %.1f ms\n", ms(tA, tB), ms(tB, tC), ms(tC, tD));
    if (bad) return BVG_E_EOF;                               // a successor outside [0,n): malformed stream
    if (!dev) {
        HIPCHK(hipMemcpy(toffsets, d_toff, (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (total) HIPCHK(hipMemcpy(tsucc, d_tsucc, (size_t)total * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

// Transform.symmetrizeOffline (Transform.java:546-575) = union(g, transposeOffline(g)): the transposition feed above, then the
// per-node sorted union of the graph's own lists (still in the transpose workspace) and the transposed ones.
static int symmetrize_impl(bvg_graph* g, uint64_t* soffsets, int64_t* ssucc, uint64_t cap, uint64_t* n_arcs, bool dev) {
    if (!g || !soffsets) return BVG_E_ARG;
    Shared* sh = g->sh;
    const int64_t n = sh->p.nodes;
    HIPCHK(hipSetDevice(sh->device));
    uint64_t arcs = 0;
    uint64_t* d_toff = nullptr; int64_t* d_ts = nullptr; int32_t* d_cnt = nullptr; uint64_t* d_soff = nullptr; uint64_t* d_tmp = nullptr; int64_t* d_out = nullptr;
    auto done = [&](int code) { for (void* p : {(void*)d_toff, (void*)d_ts, (void*)d_cnt, (void*)d_soff, (void*)d_tmp, (void*)d_out}) if (p) (void)hipFree(p); return code; };
    const size_t nn = (size_t)(n > 0 ? n : 1);
    if (hipMalloc(&d_toff, (nn + 1) * 8) != hipSuccess) return done(BVG_E_NOMEM);
    int r = transpose_impl(g, d_toff, nullptr, 0, &arcs, true);                // arc count (sum of the outdegrees)
    if (r && r != BVG_E_CAPACITY) return done(r);
    if (hipMalloc(&d_ts, (size_t)(arcs ? arcs : 1) * 8) != hipSuccess) return done(BVG_E_NOMEM);
    r = transpose_impl(g, d_toff, d_ts, arcs, &arcs, true); if (r) return done(r);
    const uint64_t* d_cum = (const uint64_t*)((char*)g->tr_ws + g->tr_o_cum); const int64_t* d_succ = (const int64_t*)((char*)g->tr_ws + g->tr_o_succ);
    if (hipMalloc(&d_cnt, nn * 4) != hipSuccess || hipMalloc(&d_soff, (nn + 1) * 8) != hipSuccess || hipMalloc(&d_tmp, scan_tmp_elems((int64_t)nn) * 8) != hipSuccess) return done(BVG_E_NOMEM);
    if (hipMemsetAsync(d_soff, 0, (nn + 1) * 8, g->stream) != hipSuccess) return done(BVG_E_HIP);
    uint64_t total = 0;
    if (n > 0) {
        launch_union_count(d_cum, d_succ, d_toff, d_ts, n, d_cnt, g->stream);
        launch_exclusive_scan(d_cnt, d_soff, n, d_tmp, g->stream);
        if (hipMemcpyAsync(&total, d_soff + n, 8, hipMemcpyDeviceToHost, g->stream) != hipSuccess) return done(BVG_E_HIP);
    }
    if (hipStreamSynchronize(g->stream) != hipSuccess) return done(BVG_E_HIP);
    if (n_arcs) *n_arcs = total;
    if (dev) { if (hipMemcpyAsync(soffsets, d_soff, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, g->stream) != hipSuccess) return done(BVG_E_HIP); }
    else if (hipMemcpy(soffsets, d_soff, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return done(BVG_E_HIP);
    if (total > cap || (!ssucc && total > 0)) { (void)hipStreamSynchronize(g->stream); return done(BVG_E_CAPACITY); }
    int64_t* d_dst = ssucc;
    if (!dev) { if (hipMalloc(&d_out, (size_t)(total ? total : 1) * 8) != hipSuccess) return done(BVG_E_NOMEM); d_dst = d_out; }
    launch_union_write(d_cum, d_succ, d_toff, d_ts, n, d_soff, d_dst, g->stream);
    if (!dev && total && hipMemcpyAsync(ssucc, d_out, (size_t)total * 8, hipMemcpyDeviceToHost, g->stream) != hipSuccess) return done(BVG_E_HIP);
    if (hipStreamSynchronize(g->stream) != hipSuccess) return done(BVG_E_HIP);
    return done(0);
}

int bvg_symmetrize(bvg_graph* g, uint64_t* soffsets, int64_t* ssucc, uint64_t ssucc_cap, uint64_t* n_arcs) {
    return symmetrize_impl(g, soffsets, ssucc, ssucc_cap, n_arcs, false);
}
int bvg_symmetrize_dev(bvg_graph* g, void* d_soffsets, void* d_ssucc, uint64_t ssucc_cap, uint64_t* n_arcs) {
    return symmetrize_impl(g, (uint64_t*)d_soffsets, (int64_t*)d_ssucc, ssucc_cap, n_arcs, true);
}

int bvg_transpose(bvg_graph* g, uint64_t* toffsets, int64_t* tsucc, uint64_t tsucc_cap, uint64_t* n_arcs) {
    return transpose_impl(g, toffsets, tsucc, tsucc_cap, n_arcs, false);
}
int bvg_transpose_dev(bvg_graph* g, void* d_toffsets, void* d_tsucc, uint64_t tsucc_cap, uint64_t* n_arcs) {
    return transpose_impl(g, (uint64_t*)d_toffsets, (int64_t*)d_tsucc, tsucc_cap, n_arcs, true);
}

// Arc-balanced split points (the skipTo() walk over algo/EliasFanoCumulativeOutdegreeList.java:30-75 that
// algo/HyperBall.java:748-768 uses for its tasks): bounds[j] = first node whose cumulative outdegree reaches j * arcs / k.
static int bvg_split_by_arcs_impl(bvg_graph* g, int k, int64_t* bounds) {
    if (!g || !bounds || k < 1) return BVG_E_ARG;
    Shared* sh = g->sh;
    HIPCHK(hipSetDevice(sh->device));
    const int64_t n = sh->p.nodes;
    if (n == 0) { for (int i = 0; i <= k; i++) bounds[i] = 0; return 0; }
    int32_t* d_deg = nullptr; uint64_t* d_cum = nullptr; uint64_t* d_tmp = nullptr; uint64_t* d_first = nullptr;
    auto done = [&](int code) { for (void* p : {(void*)d_deg, (void*)d_cum, (void*)d_tmp, (void*)d_first}) if (p) (void)hipFree(p); return code; };
    if (hipMalloc(&d_deg, (size_t)n * 4) != hipSuccess || hipMalloc(&d_cum, (size_t)(n + 1) * 8) != hipSuccess ||
        hipMalloc(&d_tmp, scan_tmp_elems(n) * 8) != hipSuccess || hipMalloc(&d_first, ((size_t)k + 1) * 8) != hipSuccess) return done(BVG_E_NOMEM);
    launch_outdegrees(sh->d_graph, sh->nbytes, sh->offs, 0, n, sh->p.outdegree_coding, d_deg, nullptr, g->stream);
    launch_exclusive_scan(d_deg, d_cum, n, d_tmp, g->stream);
    uint64_t arcs = 0;
    if (hipMemcpyAsync(&arcs, d_cum + n, 8, hipMemcpyDeviceToHost, g->stream) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) return done(BVG_E_HIP);
    uint64_t per = (arcs + (uint64_t)k - 1) / (uint64_t)k; if (per == 0) per = 1;
    launch_plan_boundaries(Offsets{nullptr, nullptr, d_cum}, n, per, (uint64_t)k, d_first, g->stream);   // (cumulative outdegrees: a plain array)
    std::vector<uint64_t> f((size_t)k + 1);
    if (hipMemcpyAsync(f.data(), d_first, ((size_t)k + 1) * 8, hipMemcpyDeviceToHost, g->stream) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) return done(BVG_E_HIP);
    for (int i = 0; i <= k; i++) bounds[i] = (int64_t)f[(size_t)i];
    bounds[0] = 0; bounds[k] = n;
    return done(0);
}

static int bvg_split_by_bits_impl(bvg_graph* g, int k, int64_t* bounds) {
    if (!g || !bounds || k < 1) return BVG_E_ARG;
    Shared* sh = g->sh;
    HIPCHK(hipSetDevice(sh->device));
    const int64_t n = sh->p.nodes;
    if (n == 0) { for (int i = 0; i <= k; i++) bounds[i] = 0; return 0; }
    uint64_t per = (sh->total_bits + (uint64_t)k - 1) / (uint64_t)k; if (per == 0) per = 1;
    uint64_t* d_first = nullptr;
    HIPCHK(hipMalloc(&d_first, ((size_t)k + 1) * sizeof(uint64_t)));
    launch_plan_boundaries(sh->offs, n, per, (uint64_t)k, d_first, g->stream);
    std::vector<uint64_t> f((size_t)k + 1);
    hipError_t e = hipMemcpyAsync(f.data(), d_first, ((size_t)k + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    (void)hipFree(d_first);
    if (e != hipSuccess) return BVG_E_HIP;
    for (int i = 0; i <= k; i++) bounds[i] = (int64_t)f[(size_t)i];
    bounds[0] = 0; bounds[k] = n;
    return 0;
}

// Shard bounds of a k-way split of the node range (ImmutableGraph.splitNodeIterators, IG:405-436): BVG_BALANCE_NODES is the
// reference's own rule (ceil(n/k) nodes each, IG:415-433), BVG_BALANCE_BITS / _ARCS the balanced variants above.  Cached per
// (k, balance) in the shared part of the handle, so every flyweight and every later call agrees on them.
static int bvg_shard_bounds_impl(bvg_graph* g, int k, int balance, int64_t* bounds) {
    if (!g || !bounds || k < 1 || balance < BVG_BALANCE_NODES || balance > BVG_BALANCE_ARCS) return BVG_E_ARG;
    Shared* sh = g->sh;
    const uint64_t key = ((uint64_t)k << 2) | (uint64_t)balance;
    std::lock_guard<std::mutex> lk(sh->shard_mu);
    auto it = sh->shard_bounds.find(key);
    if (it == sh->shard_bounds.end()) {
        std::vector<int64_t> b((size_t)k + 1);
        int r = 0;
        if (balance == BVG_BALANCE_NODES) {
            const int64_t n = sh->p.nodes, m = n ? (n + k - 1) / k : 0;
            for (int i = 0; i <= k; i++) b[(size_t)i] = std::min<int64_t>((int64_t)i * m, n);
        } else r = balance == BVG_BALANCE_BITS ? bvg_split_by_bits_impl(g, k, b.data()) : bvg_split_by_arcs_impl(g, k, b.data());
        if (r) return r;
        it = sh->shard_bounds.emplace(key, std::move(b)).first;
    }
    memcpy(bounds, it->second.data(), ((size_t)k + 1) * sizeof(int64_t));
    return 0;
}

// Shard r of k: the scan of nodes [bounds[r], bounds[r+1]) (one rank of the multi-GPU scan; the caller reduces {arcs, chk}).
static int bvg_scan_shard_impl(bvg_graph* g, int k, int r, int balance, bvg_scan_result* out, int64_t* from, int64_t* to) {
    if (!g || !out || k < 1 || r < 0 || r >= k) return BVG_E_ARG;
    std::vector<int64_t> b((size_t)k + 1);
    int rc = bvg_shard_bounds_impl(g, k, balance, b.data()); if (rc) return rc;
    if (from) *from = b[(size_t)r];
    if (to) *to = b[(size_t)r + 1];
    return bvg_scan_impl(g, b[(size_t)r], b[(size_t)r + 1], out);
}

// One process, several GPUs (what a JVM host has): handle i scans shard i of ngpu on its own device, all at once (one host
// thread each), and the per-shard {nodes, arcs, chk} are summed on the host -- 24 bytes, so no device collective is involved
// (the one-process-per-GPU form of the same reduction is bench.py's RCCL all-reduce).  The handles must describe the same graph
// (replicas opened on different devices, or bvg_copy() flyweights on one device) and be DISTINCT: a handle owns one stream and
// one set of result words.  kernel_ms = the slowest shard.
static int bvg_scan_multi_impl(bvg_graph* const* per_gpu, int ngpu, int balance, bvg_scan_result* total, bvg_scan_result* per_shard) {
    if (!per_gpu || ngpu < 1 || !total) return BVG_E_ARG;
    for (int i = 0; i < ngpu; i++) {
        if (!per_gpu[i]) return BVG_E_ARG;
        for (int j = 0; j < i; j++) if (per_gpu[j] == per_gpu[i]) return BVG_E_ARG;              // two threads on one handle would share its stream and results
        const bvg_params &a = per_gpu[0]->sh->p, &b = per_gpu[i]->sh->p;
        if (a.nodes != b.nodes || a.arcs != b.arcs || per_gpu[0]->sh->total_bits != per_gpu[i]->sh->total_bits) return BVG_E_ARG;
    }
    std::vector<int64_t> bounds((size_t)ngpu + 1);
    int rc = bvg_shard_bounds_impl(per_gpu[0], ngpu, balance, bounds.data()); if (rc) return rc;
    std::vector<bvg_scan_result> res((size_t)ngpu);
    std::vector<int> st((size_t)ngpu, BVG_E_HIP);
    std::vector<std::thread> th;
    th.reserve((size_t)ngpu);
    struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};   // also when emplace_back throws
    for (int i = 0; i < ngpu; i++)
        th.emplace_back([&, i] { st[(size_t)i] = guarded([&] { return bvg_scan_impl(per_gpu[i], bounds[(size_t)i], bounds[(size_t)i + 1], &res[(size_t)i]); }); });
    for (auto& t : th) t.join();
    memset(total, 0, sizeof *total);
    for (int i = 0; i < ngpu; i++) {
        if (st[(size_t)i]) return st[(size_t)i];
        const bvg_scan_result& x = res[(size_t)i];
        total->nodes += x.nodes; total->arcs += x.arcs; total->chk += x.chk; total->graph_bytes += x.graph_bytes; total->index_bytes += x.index_bytes;
        total->launches += x.launches; total->slow_blocks += x.slow_blocks; total->index_entries += x.index_entries; total->lean_blocks += x.lean_blocks;
        if (x.kernel_ms > total->kernel_ms) total->kernel_ms = x.kernel_ms;
        if (per_shard) per_shard[i] = x;
    }
    return 0;
}

// the skip index (and the validation that comes with it) for the blocks of [from, to), built now
static int bvg_build_index_impl(bvg_graph* g, int64_t from, int64_t to, uint64_t* entries, uint64_t* bytes) {
    if (!g) return BVG_E_ARG;
    Shared* sh = g->sh;
    if (from < 0 || to > sh->p.nodes || from > to) return BVG_E_ARG;
    HIPCHK(hipSetDevice(sh->device));
    if (entries) *entries = 0;
    if (bytes) *bytes = 0;
    const bool rows_ok = (g->tun.reserved & 0xFF) == 0 && !g->tun.force_slow && sh->p.window_size <= kMaxWindow;
    if (!rows_ok || from == to) return 0;                                   // the global-memory kernels use no index
    std::shared_ptr<Plan> plp;
    int r = build_plan(g, block_bits_of(g), plp); if (r) return r;
    const std::vector<uint64_t>& hf = plp->h_first;
    uint32_t lo = (uint32_t)(std::upper_bound(hf.begin(), hf.end(), (uint64_t)from) - hf.begin()); lo = lo ? lo - 1 : 0;
    uint32_t hi = (uint32_t)(std::lower_bound(hf.begin(), hf.end(), (uint64_t)to) - hf.begin()); if (hi > plp->nblk) hi = plp->nblk;
    if (hi > lo) { r = build_skip(g, plp, lo, hi, /*retry_failed=*/true); if (r) return r; }
    std::shared_ptr<SkipIndex> ix = std::atomic_load(&plp->skip);
    if (ix) {
        if (entries) *entries = ix->total;
        if (bytes) *bytes = ix->total * (ix->wide ? 10u : 6u) + (uint64_t)plp->nblk * 9u;
    }
    return 0;
}

static int bvg_mosaic_impl(const bvg_graph* const* bases, int k, int64_t cycles, bvg_graph** out) {
    if (!bases || !out || k < 1 || k > kMosaicMax || cycles < 1) return BVG_E_ARG;
    for (int i = 0; i < k; i++) if (!bases[i]) return BVG_E_ARG;
    Shared* b0 = bases[0]->sh;
    HIPCHK(hipSetDevice(b0->device));
    MosaicSrc m{}; m.k = k;
    int64_t arcs = 0;
    for (int i = 0; i < k; i++) {
        Shared* b = bases[i]->sh;
        const bvg_params &p = b->p, &q = b0->p;
        if (b->device != b0->device || b->p.nodes <= 0 || b->total_bits == 0) return BVG_E_ARG;
        if (p.window_size != q.window_size || p.min_interval_length != q.min_interval_length || p.zeta_k != q.zeta_k || p.outdegree_coding != q.outdegree_coding ||
            p.block_coding != q.block_coding || p.residual_coding != q.residual_coding || p.reference_coding != q.reference_coding || p.block_count_coding != q.block_count_coding) return BVG_E_ARG;
        m.graph[i] = b->d_graph; m.offs[i] = b->offs; m.bits[i] = b->total_bits;
        m.bit_prefix[i + 1] = m.bit_prefix[i] + b->total_bits; m.node_prefix[i + 1] = m.node_prefix[i] + p.nodes;
        arcs = (arcs < 0 || p.arcs < 0) ? -1 : arcs + p.arcs;
    }
    m.cycle_bits = m.bit_prefix[k]; m.cycle_nodes = m.node_prefix[k];
    if (cycles > INT64_MAX / m.cycle_nodes) return BVG_E_ARG;
    const uint64_t total_bits = m.cycle_bits * (uint64_t)cycles;
    const uint64_t nbytes = (total_bits + 7) / 8;
    const uint64_t padded = ((nbytes + 15) & ~15ull) + kPad;
    const int64_t n = m.cycle_nodes * cycles;
    uint8_t* d_graph = nullptr; uint32_t* d_lo = nullptr; uint64_t* d_hi = nullptr;
    const size_t n1 = (size_t)n + 1;
    DevBuf flag;
    if (flag.alloc(sizeof(unsigned)) || hipMemset(flag.p, 0, sizeof(unsigned)) != hipSuccess) return BVG_E_NOMEM;
    HIPCHK(hipMalloc(&d_graph, padded));
    if (hipMalloc(&d_lo, n1 * sizeof(uint32_t)) != hipSuccess || hipMalloc(&d_hi, ((n1 >> kOffShift) + 2) * sizeof(uint64_t)) != hipSuccess) {
        (void)hipFree(d_graph); if (d_lo) (void)hipFree(d_lo); return BVG_E_NOMEM;
    }
    hipStream_t st = bases[0]->stream;
    launch_mosaic_graph(m, d_graph, padded, cycles, st);
    launch_mosaic_offsets(m, cycles, d_lo, d_hi, (unsigned*)flag.p, st);
    unsigned over = 0;
    if (hipMemcpyAsync(&over, flag.p, sizeof over, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) over = 2;
    if (over) { (void)hipFree(d_graph); (void)hipFree(d_lo); (void)hipFree(d_hi); return over == 2 ? BVG_E_HIP : BVG_E_UNSUPPORTED; }   // a group of 2^kOffShift records spans 2^32 bits: not packable
    bvg_params p = b0->p; p.nodes = n; p.arcs = arcs < 0 ? -1 : arcs * cycles;
    const PackedOffsets pk{d_lo, d_hi};
    int r = open_common(&p, nullptr, d_graph, nbytes, nullptr, nullptr, b0->device, out, &pk);
    if (r) { (void)hipFree(d_graph); return r; }                      // (open_common owns the index from the start)
    (*out)->sh->own_graph = true;
    (*out)->tun = bases[0]->tun;
    return 0;
}
static int bvg_tile_impl(const bvg_graph* base, int64_t copies, bvg_graph** out) { return bvg_mosaic_impl(&base, 1, copies, out); }

int bvg_scan(bvg_graph* g, int64_t from, int64_t to, bvg_scan_result* out) { return guarded([&] { return bvg_scan_impl(g, from, to, out); }); }
// Shard bounds of a k-way split of the node range (ImmutableGraph.splitNodeIterators, IG:405-436): BVG_BALANCE_NODES is the
// reference's own rule (ceil(n/k) nodes each, IG:415-433), BVG_BALANCE_BITS / _ARCS the balanced variants.  Cached per
// (k, balance) in the shared part of the handle, so every flyweight and every later call agrees on them.
int bvg_shard_bounds(bvg_graph* g, int k, int balance, int64_t* bounds) { return guarded([&] { return bvg_shard_bounds_impl(g, k, balance, bounds); }); }
int bvg_scan_shard(bvg_graph* g, int k, int r, int balance, bvg_scan_result* out, int64_t* from, int64_t* to) { return guarded([&] { return bvg_scan_shard_impl(g, k, r, balance, out, from, to); }); }
int bvg_scan_multi(bvg_graph* const* per_gpu, int ngpu, int balance, bvg_scan_result* total, bvg_scan_result* per_shard) { return guarded([&] { return bvg_scan_multi_impl(per_gpu, ngpu, balance, total, per_shard); }); }
int bvg_build_index(bvg_graph* g, int64_t from, int64_t to, uint64_t* entries, uint64_t* bytes) { return guarded([&] { return bvg_build_index_impl(g, from, to, entries, bytes); }); }
int bvg_save_index(bvg_graph* g, const char* path) { return guarded([&] { return save_index_impl(g, path); }); }
int bvg_load_index(bvg_graph* g, const char* path) { return guarded([&] { return load_index_impl(g, path); }); }
int bvg_successors_batch(bvg_graph* g, const int64_t* nodes, int64_t count, int32_t* outdeg, int64_t* succ, uint64_t succ_cap, uint64_t* n_succ) { return guarded([&] { return bvg_successors_batch_impl(g, nodes, count, outdeg, succ, succ_cap, n_succ); }); }
int bvg_tile(const bvg_graph* base, int64_t copies, bvg_graph** out) { return guarded([&] { return bvg_tile_impl(base, copies, out); }); }
int bvg_mosaic(const bvg_graph* const* bases, int k, int64_t cycles, bvg_graph** out) { return guarded([&] { return bvg_mosaic_impl(bases, k, cycles, out); }); }
int bvg_split_by_arcs(bvg_graph* g, int k, int64_t* bounds) { return guarded([&] { return bvg_split_by_arcs_impl(g, k, bounds); }); }
int bvg_split_by_bits(bvg_graph* g, int k, int64_t* bounds) { return guarded([&] { return bvg_split_by_bits_impl(g, k, bounds); }); }

// BVGraph.store(graph, basename, W, maxRef, minInterval, zetaK, flags) on the device (BVG:2329-2470 around CompressionThread.call,
// :2216-2327): see bvg_encode.hip.  Host buffers in, malloc'ed host buffers out.
int bvg_store(const bvg_params* p, int64_t nodes, const uint64_t* adj_off, const int64_t* adj, int64_t chunk_nodes, int device,
              uint8_t** graph, uint64_t* graph_bytes, uint64_t** offsets) {
    if (!p || !adj_off || !graph || !graph_bytes || !offsets || nodes < 0) return BVG_E_ARG;
    return guarded([&]() -> int {
        bvg_params q = *p; q.nodes = nodes;
        int r = check_params(q); if (r) return r;
        r = ensure_device(device); if (r) return r;
        const uint64_t m = adj_off[nodes];
        if (m && !adj) return BVG_E_ARG;
        uint64_t* d_off = nullptr; int64_t* d_adj = nullptr; uint8_t* d_graph = nullptr; uint64_t* d_offsets = nullptr;
        auto done = [&](int code) { for (void* x : {(void*)d_off, (void*)d_adj, (void*)d_graph, (void*)d_offsets}) if (x) (void)hipFree(x); return code; };
        if (hipMalloc(&d_off, ((size_t)nodes + 1) * sizeof(uint64_t)) != hipSuccess || hipMalloc(&d_adj, (size_t)(m ? m : 1) * sizeof(int64_t)) != hipSuccess) { (void)hipGetLastError(); return done(BVG_E_NOMEM); }
        if (hipMemcpy(d_off, adj_off, ((size_t)nodes + 1) * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return done(BVG_E_HIP);
        if (m && hipMemcpy(d_adj, adj, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) return done(BVG_E_HIP);
        uint64_t nbytes = 0;
        r = encode_store_dev(q, d_off, d_adj, nodes, chunk_nodes, nullptr, &d_graph, &nbytes, &d_offsets);
        if (r) return done(r);
        uint8_t* hg = (uint8_t*)calloc((size_t)nbytes + 16, 1); uint64_t* ho = (uint64_t*)malloc(((size_t)nodes + 1) * sizeof(uint64_t));
        if (!hg || !ho) { free(hg); free(ho); return done(BVG_E_NOMEM); }
        if ((nbytes && hipMemcpy(hg, d_graph, (size_t)nbytes, hipMemcpyDeviceToHost) != hipSuccess) ||
            hipMemcpy(ho, d_offsets, ((size_t)nodes + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) { free(hg); free(ho); return done(BVG_E_HIP); }
        *graph = hg; *graph_bytes = nbytes; *offsets = ho;
        return done(0);
    });
}
void bvg_free(void* p) { free(p); }

uint64_t bvg_arc_mix(uint64_t x, uint64_t y) {
    uint32_t k0, k1; node_key(x, k0, k1);
    return mix_keyed(k0, k1, y);
}

const char* bvg_strerror(int status) {
    switch (status) {
        case BVG_OK: return "ok";
        case BVG_E_ARG: return "argument out of range (IllegalArgumentException)";
        case BVG_E_STATE: return "illegal state: reference beyond window, or offsets unavailable (IllegalStateException)";
        case BVG_E_UNSUPPORTED: return "unsupported coding / window / access mode (UnsupportedOperationException)";
        case BVG_E_IO: return "cannot read or parse basename.{properties,graph,offsets} (IOException)";
        case BVG_E_EOF: return "bit stream inconsistent with offsets or truncated (EOFException)";
        case BVG_E_NOMEM: return "out of host or device memory";
        case BVG_E_HIP: return "no gfx950 device or HIP runtime failure (this library has no CPU fallback)";
        case BVG_E_CAPACITY: return "successor buffer too small";
    }
    return "unknown status";
}

}  // extern "C"
