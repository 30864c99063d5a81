// bvg_store.cpp — CPU BVGraph *encoder* and synthetic-graph generators (test / bench tooling).
//
// Not part of the GPU hot path: it exists so that tests and bench.py can manufacture .graph bit
// streams (synthetic web-like graphs, the window=0 re-store of BASELINE config 2, micro-graphs for
// every decoder branch) without a JVM.  It restates the reference's compressor
//   BVGraph.java:1595-1618 (intervalize), :1977-2159 (diffComp), :2216-2327 (CompressionThread.call)
// and is held to a bit-exact standard: tests/test_store.py regenerates cnr-2000.graph and
// cnr-2000.offsets byte-for-byte from the text golden.
//
// Output of a "chunked" store (chunk_nodes > 0) mirrors the reference's multi-threaded store
// (BVGraph.java:2404-2457): every chunk is compressed with a fresh window and the per-chunk bit
// streams are concatenated bit-wise, so the result does not depend on the number of threads.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../include/bvgraph_hip.h"

namespace {

inline int msb64(uint64_t x) { return 63 - __builtin_clzll(x); }
inline uint64_t int2nat(int64_t v) { return v >= 0 ? (uint64_t)v << 1 : (((uint64_t)(-(v + 1))) << 1) + 1; }  // Fast.int2nat

// ---- bit sinks: a real MSB-first writer and a counter (the reference's NullOutputStream dry run) ----
struct BitWriter {
    std::vector<uint8_t> bytes;
    uint64_t acc = 0; int nacc = 0; uint64_t nbits = 0;
    void put(uint64_t v, int n) {  // n <= 64 low bits of v, MSB first
        nbits += (uint64_t)n;
        while (n > 0) {
            int take = std::min(n, 56 - nacc);             // nacc < 8 between calls, so take >= 1
            uint64_t piece = (v >> (n - take)) & ((1ULL << take) - 1);
            acc = (acc << take) | piece; nacc += take; n -= take;
            while (nacc >= 8) { bytes.push_back((uint8_t)(acc >> (nacc - 8))); nacc -= 8; }
            acc &= (1ULL << nacc) - 1;
        }
    }
    void zeros(uint64_t n) { while (n >= 32) { put(0, 32); n -= 32; } if (n) put(0, (int)n); }
    void flush() { if (nacc) { bytes.push_back((uint8_t)(acc << (8 - nacc))); nacc = 0; } }
};
struct BitCounter {
    uint64_t nbits = 0;
    void put(uint64_t, int n) { nbits += (uint64_t)n; }
    void zeros(uint64_t n) { nbits += n; }
};

// dsiutils OutputBitStream codes
template <class S> void write_unary(S& s, uint64_t x) { s.zeros(x); s.put(1, 1); }
template <class S> void write_gamma(S& s, uint64_t x) { int b = msb64(x + 1); write_unary(s, (uint64_t)b); if (b) s.put((x + 1) & ((1ULL << b) - 1), b); }
template <class S> void write_delta(S& s, uint64_t x) { int b = msb64(x + 1); write_gamma(s, (uint64_t)b); if (b) s.put((x + 1) & ((1ULL << b) - 1), b); }
template <class S> void write_zeta(S& s, uint64_t x, int k) {
    uint64_t v = x + 1; int h = msb64(v) / k; uint64_t left = 1ULL << (h * k);
    write_unary(s, (uint64_t)h);
    if (v - left < left) s.put(v - left, h * k + k - 1);
    else s.put(v, h * k + k);
}
template <class S> void write_nibble(S& s, uint64_t x) {
    if (x == 0) { s.put(8, 4); return; }
    int h = msb64(x) / 3;
    do { s.put(h == 0, 1); s.put((x >> (h * 3)) & 7, 3); } while (h-- != 0);
}
template <class S> void write_golomb(S& s, uint64_t x, uint64_t b) {
    if (b == 0) return;
    write_unary(s, x / b);
    if (b == 1) return;
    int l = msb64(b); uint64_t thr = (1ULL << (l + 1)) - b, r = x % b;
    if (r < thr) s.put(r, l); else s.put(r + thr, l + 1);
}
template <class S> void write_coded(S& s, uint64_t x, int coding, int k) {
    switch (coding) {
        case BVG_GAMMA: write_gamma(s, x); break;
        case BVG_DELTA: write_delta(s, x); break;
        case BVG_UNARY: write_unary(s, x); break;
        case BVG_ZETA: write_zeta(s, x, k); break;
        case BVG_NIBBLE: write_nibble(s, x); break;
        case BVG_GOLOMB: write_golomb(s, x, (uint64_t)k); break;
        default: write_gamma(s, x);
    }
}

struct Stats { uint64_t copied = 0, intervalised = 0, residual = 0, tot_ref = 0, tot_dist = 0, nodes_with_ref = 0; };

struct Compressor {
    const bvg_params& p;
    std::vector<int64_t> extras, left, len, residuals; std::vector<int32_t> blocks;
    explicit Compressor(const bvg_params& pp) : p(pp) {}

    // BVGraph.intervalize, BVGraph.java:1595-1618
    int intervalize(const std::vector<int64_t>& x, int min_interval) {
        int n_interval = 0; const int64_t vl = (int64_t)x.size(); const int64_t* v = x.data();
        left.clear(); len.clear(); residuals.clear();
        for (int64_t i = 0; i < vl; i++) {
            int64_t j = 0;
            if (i < vl - 1 && v[i] + 1 == v[i + 1]) {
                do j++; while (i + j < vl - 1 && v[i + j] + 1 == v[i + j + 1]);
                j++;
                if (j >= min_interval) { left.push_back(v[i]); len.push_back(j); n_interval++; i += j - 1; }
            }
            if (j < min_interval) residuals.push_back(v[i]);
        }
        return n_interval;
    }

    // CompressionThread.diffComp, BVGraph.java:1977-2159
    template <class S>
    void diff_comp(S& obs, int64_t curr_node, int ref, const int64_t* ref_list, int64_t ref_len, const int64_t* curr_list, int64_t curr_len, Stats* st) {
        int64_t j = 0, k = 0; int32_t curr_block_len = 0; bool copying = true;
        if (ref == 0) ref_len = 0;
        extras.clear(); blocks.clear();
        while (j < curr_len && k < ref_len) {
            if (copying) {
                if (curr_list[j] > ref_list[k]) { blocks.push_back(curr_block_len); copying = false; curr_block_len = 0; }
                else if (curr_list[j] < ref_list[k]) extras.push_back(curr_list[j++]);
                else { j++; k++; curr_block_len++; if (st) st->copied++; }
            } else {
                if (curr_list[j] < ref_list[k]) extras.push_back(curr_list[j++]);
                else if (curr_list[j] > ref_list[k]) { k++; curr_block_len++; }
                else { blocks.push_back(curr_block_len); copying = true; curr_block_len = 0; }
            }
        }
        if (copying && k < ref_len) blocks.push_back(curr_block_len);
        while (j < curr_len) extras.push_back(curr_list[j++]);

        if (p.window_size > 0) write_coded(obs, (uint64_t)ref, p.reference_coding, 0);
        if (ref != 0) {
            write_coded(obs, (uint64_t)blocks.size(), p.block_count_coding, 0);
            if (!blocks.empty()) {
                write_coded(obs, (uint64_t)blocks[0], p.block_coding, 0);
                for (size_t i = 1; i < blocks.size(); i++) write_coded(obs, (uint64_t)(blocks[i] - 1), p.block_coding, 0);
            }
        }
        if (!extras.empty()) {
            const std::vector<int64_t>* res = &extras;
            if (p.min_interval_length != 0) {
                int ic = intervalize(extras, p.min_interval_length);
                write_gamma(obs, (uint64_t)ic);
                int64_t prev = 0;
                for (int i = 0; i < ic; i++) {
                    if (i == 0) write_gamma(obs, int2nat((prev = left[0]) - curr_node));
                    else write_gamma(obs, (uint64_t)(left[i] - prev - 1));
                    prev = left[i] + len[i];
                    if (st) st->intervalised += (uint64_t)len[i];
                    write_gamma(obs, (uint64_t)(len[i] - p.min_interval_length));
                }
                res = &residuals;
            }
            if (!res->empty()) {
                if (st) st->residual += res->size();
                int64_t prev = (*res)[0];
                write_coded(obs, int2nat(prev - curr_node), p.residual_coding, p.zeta_k);
                for (size_t i = 1; i < res->size(); i++) {
                    write_coded(obs, (uint64_t)((*res)[i] - prev - 1), p.residual_coding, p.zeta_k);
                    prev = (*res)[i];
                }
            }
        }
    }
};

// Source of successor lists for a node range: either an in-memory adjacency or a generator.
struct ListSource {
    virtual ~ListSource() {}
    // Called for x = first, first+1, ... in order within a chunk; fills out (sorted, unique).
    virtual void begin_chunk(int64_t first) = 0;
    virtual void next(int64_t x, std::vector<int64_t>& out) = 0;
};

struct ChunkOut { BitWriter g; std::vector<uint64_t> node_bits; Stats st; uint64_t arcs = 0; };

// CompressionThread.call, BVGraph.java:2163-2327, for nodes [first, last).
void compress_range(const bvg_params& p, ListSource& src, int64_t first, int64_t last, ChunkOut& out) {
    const int W = p.window_size, cyc = W + 1;
    const int64_t max_ref = p.max_ref_count < 0 ? INT64_MAX : p.max_ref_count;   // -m -1 == unbounded (BVGraph.java:2654)
    std::vector<std::vector<int64_t>> list((size_t)cyc);
    std::vector<int64_t> ref_count((size_t)cyc, 0);
    Compressor c(p); BitCounter bc;
    src.begin_chunk(first);
    out.node_bits.reserve((size_t)(last - first));
    for (int64_t x = first; x < last; x++) {
        const int ci = (int)((x - first) % cyc);      // window restarts at the chunk start, as per-thread stores do
        uint64_t start = out.g.nbits;
        std::vector<int64_t>& cur = list[(size_t)ci];
        src.next(x, cur);
        const int64_t d = (int64_t)cur.size();
        write_coded(out.g, (uint64_t)d, p.outdegree_coding, 0);
        if (d > 0) {
            uint64_t best = UINT64_MAX; int best_cand = -1, best_ref = -1;
            ref_count[(size_t)ci] = -1;
            for (int ref = 0; ref < cyc; ref++) {
                if (x - ref < first && ref != 0) break;            // nothing before the chunk start
                int cand = (int)(((x - first) - ref + 2LL * cyc) % cyc);
                if (ref_count[(size_t)cand] < max_ref && !list[(size_t)cand].empty()) {
                    bc.nbits = 0;
                    c.diff_comp(bc, x, ref, list[(size_t)cand].data(), (int64_t)list[(size_t)cand].size(), cur.data(), d, nullptr);
                    if (bc.nbits < best) { best = bc.nbits; best_cand = cand; best_ref = ref; }
                }
            }
            ref_count[(size_t)ci] = ref_count[(size_t)best_cand] + 1;
            c.diff_comp(out.g, x, best_ref, list[(size_t)best_cand].data(), (int64_t)list[(size_t)best_cand].size(), cur.data(), d, &out.st);
            out.st.tot_ref += (uint64_t)ref_count[(size_t)ci]; out.st.tot_dist += (uint64_t)best_ref;
            if (best_ref) out.st.nodes_with_ref++;
            out.arcs += (uint64_t)d;
        }
        out.node_bits.push_back(out.g.nbits - start);
    }
}

struct AdjSource : ListSource {
    const uint64_t* off; const int64_t* adj;
    AdjSource(const uint64_t* o, const int64_t* a) : off(o), adj(a) {}
    void begin_chunk(int64_t) override {}
    void next(int64_t x, std::vector<int64_t>& out) override { out.assign(adj + off[x], adj + off[x + 1]); }
};

// ---- xoroshiro128+ (the generator family SpeedTest uses, test/SpeedTest.java:75) ----
struct Rng {
    uint64_t s0, s1;
    static uint64_t sm(uint64_t& z) { z += 0x9E3779B97F4A7C15ULL; uint64_t r = z; r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ULL; r = (r ^ (r >> 27)) * 0x94D049BB133111EBULL; return r ^ (r >> 31); }
    explicit Rng(uint64_t seed) { uint64_t z = seed; s0 = sm(z); s1 = sm(z); }
    uint64_t next() { uint64_t a = s0, b = s1, r = a + b; b ^= a; s0 = ((a << 24) | (a >> 40)) ^ b ^ (b << 16); s1 = (b << 37) | (b >> 27); return r; }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    uint64_t geometric(double mean) { if (mean <= 0) return 0; double u = unit(); return (uint64_t)std::floor(std::log(1.0 - u) / std::log(mean / (mean + 1.0))); }
};

// Web-like synthetic graph ("copy model" with locality), SURVEY 8(d) config 3 stand-in.
// Each chunk of chunk_nodes nodes is generated from (seed, chunk index) alone.
struct SynthParams {
    double p_empty;      // fraction of nodes with outdegree 0
    double mean_deg;     // mean outdegree of non-empty nodes (heavy tailed)
    double tail_alpha;   // Pareto shape for the degree tail
    int64_t max_deg;
    double p_copy;       // probability that a node copies from a node <= W back
    double keep_run, skip_run;  // mean lengths of kept / skipped runs of the reference list
    double p_interval;   // probability of adding a run of consecutive successors
    double interval_len; // mean interval length (beyond 4)
    double local_gap;    // mean gap between local residuals
    double p_far;        // fraction of residuals that are uniformly random ("far" links)
    int32_t window;      // how far back copies reach
    double extra_mean;   // mean number of non-copied successors of a copying node
};

struct SynthSource : ListSource {
    SynthParams sp; int64_t n; uint64_t seed; int64_t chunk_nodes;
    std::vector<std::vector<int64_t>> hist; Rng rng{0}; int64_t chunk_first = 0;
    SynthSource(const SynthParams& s, int64_t nn, uint64_t sd, int64_t cn) : sp(s), n(nn), seed(sd), chunk_nodes(cn) { hist.resize((size_t)sp.window + 1); }
    void begin_chunk(int64_t first) override {
        chunk_first = first;
        rng = Rng(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(chunk_nodes ? first / chunk_nodes : 0) + 1);
        for (auto& h : hist) h.clear();
    }
    void next(int64_t x, std::vector<int64_t>& out) override {
        out.clear();
        const size_t slot = (size_t)((x - chunk_first) % (sp.window + 1));
        if (rng.unit() >= sp.p_empty) {
            // heavy-tailed target degree with the requested mean: Pareto(alpha) scaled
            double a = sp.tail_alpha, xm = sp.mean_deg * (a - 1.0) / a;
            double dd = xm / std::pow(1.0 - rng.unit(), 1.0 / a);
            int64_t d = (int64_t)std::llround(dd); if (d < 1) d = 1; if (d > sp.max_deg) d = sp.max_deg; if (d > n) d = n;
            bool copied_any = false;
            if (sp.window > 0 && rng.unit() < sp.p_copy) {
                int r = 1 + (int)std::min<uint64_t>(rng.geometric(1.2), (uint64_t)sp.window - 1);
                if (x - r >= chunk_first) {
                    const std::vector<int64_t>& ref = hist[(size_t)((x - r - chunk_first) % (sp.window + 1))];
                    size_t i = 0; bool keep = rng.unit() < 0.8;
                    while (i < ref.size()) {
                        uint64_t run = 1 + rng.geometric(keep ? sp.keep_run : sp.skip_run);
                        if (keep) for (uint64_t t = 0; t < run && i < ref.size(); t++) out.push_back(ref[i++]);
                        else i += run;
                        keep = !keep;
                    }
                    copied_any = !out.empty();
                }
            }
            // a node that copies is "a page of the same site": a few extra links on top of the copied ones
            if (copied_any) d = (int64_t)out.size() + (int64_t)rng.geometric(sp.extra_mean);
            int64_t missing = d - (int64_t)out.size();
            if (missing > 0 && rng.unit() < sp.p_interval) {
                int64_t l = 4 + (int64_t)rng.geometric(sp.interval_len); if (l > missing) l = missing;
                int64_t start = x + bvg_nat_shift((int64_t)rng.geometric(30.0));
                if (start < 0) start = 0;
                if (start + l > n) start = n - l;
                if (start < 0) { start = 0; l = n; }
                for (int64_t t = 0; t < l; t++) out.push_back(start + t);
                missing -= l;
            }
            if (missing > 0) {
                int64_t cur = x + bvg_nat_shift((int64_t)rng.geometric(sp.local_gap * 4));
                for (int64_t t = 0; t < missing; t++) {
                    if (rng.unit() < sp.p_far) out.push_back((int64_t)rng.below((uint64_t)n));
                    else { if (cur < 0) cur = 0; if (cur >= n) cur = (int64_t)rng.below((uint64_t)n); out.push_back(cur); cur += 1 + (int64_t)rng.geometric(sp.local_gap); }
                }
            }
            std::sort(out.begin(), out.end());
            out.erase(std::unique(out.begin(), out.end()), out.end());
        }
        hist[slot] = out;
    }
    int64_t bvg_nat_shift(int64_t g) { return (rng.next() & 1) ? g : -g; }
};

struct StoreOut {
    uint8_t* graph = nullptr; uint64_t graph_bits = 0; uint64_t graph_bytes = 0;
    uint64_t* offsets = nullptr; uint64_t arcs = 0; Stats st;
};

// Appends `bits` bits of src (MSB-first stream) to w.
void append_bits(BitWriter& w, const std::vector<uint8_t>& src, uint64_t bits) {
    uint64_t full = bits / 8;
    if (w.nacc == 0) { w.bytes.insert(w.bytes.end(), src.begin(), src.begin() + (ptrdiff_t)full); w.nbits += full * 8; }
    else for (uint64_t i = 0; i < full; i++) w.put(src[i], 8);
    int rem = (int)(bits & 7);
    if (rem) w.put((uint64_t)(src[full] >> (8 - rem)), rem);
}

template <class MakeSource>
int store_generic(const bvg_params& p, int64_t n, int64_t chunk_nodes, int nthreads, MakeSource make_source, StoreOut& out) {
    if (chunk_nodes <= 0 || chunk_nodes > n) chunk_nodes = n > 0 ? n : 1;
    const int64_t nchunks = n ? (n + chunk_nodes - 1) / chunk_nodes : 0;
    std::vector<ChunkOut> chunks((size_t)nchunks);
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back([&, t]() {
        auto src = make_source();
        for (int64_t c = t; c < nchunks; c += nthreads) {
            int64_t first = c * chunk_nodes, last = std::min(n, first + chunk_nodes);
            compress_range(p, *src, first, last, chunks[(size_t)c]);
            chunks[(size_t)c].g.flush();
        }
    });
    for (auto& t : th) t.join();
    uint64_t total_bits = 0;
    for (auto& c : chunks) total_bits += c.g.nbits;
    BitWriter w; w.bytes.reserve((size_t)(total_bits / 8 + 64));
    out.offsets = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    if (!out.offsets) return BVG_E_NOMEM;
    uint64_t pos = 0; int64_t x = 0;
    for (auto& c : chunks) {
        for (uint64_t b : c.node_bits) { out.offsets[x++] = pos; pos += b; }
        append_bits(w, c.g.bytes, c.g.nbits);
        out.arcs += c.arcs;
        out.st.copied += c.st.copied; out.st.intervalised += c.st.intervalised; out.st.residual += c.st.residual;
        out.st.tot_ref += c.st.tot_ref; out.st.tot_dist += c.st.tot_dist; out.st.nodes_with_ref += c.st.nodes_with_ref;
        std::vector<uint8_t>().swap(c.g.bytes);
    }
    out.offsets[n] = pos;
    w.flush();
    out.graph_bits = pos; out.graph_bytes = w.bytes.size();
    out.graph = (uint8_t*)malloc(w.bytes.size() + 64);
    if (!out.graph) return BVG_E_NOMEM;
    memcpy(out.graph, w.bytes.data(), w.bytes.size()); memset(out.graph + w.bytes.size(), 0, 64);
    return 0;
}

}  // namespace

extern "C" {

struct bvgt_stats { uint64_t arcs, copied, intervalised, residual, tot_ref, tot_dist, nodes_with_ref, graph_bits, graph_bytes; };

struct bvgt_synth_params {
    double p_empty, mean_deg, tail_alpha; int64_t max_deg;
    double p_copy, keep_run, skip_run, p_interval, interval_len, local_gap, p_far; int32_t window; int32_t pad; double extra_mean;
};

static void fill_stats(const StoreOut& o, bvgt_stats* s) {
    if (!s) return;
    s->arcs = o.arcs; s->copied = o.st.copied; s->intervalised = o.st.intervalised; s->residual = o.st.residual;
    s->tot_ref = o.st.tot_ref; s->tot_dist = o.st.tot_dist; s->nodes_with_ref = o.st.nodes_with_ref;
    s->graph_bits = o.graph_bits; s->graph_bytes = o.graph_bytes;
}

// BVGraph.store(graph, basename, W, maxRef, minInterval, zetaK, flags) over an in-memory adjacency
// (adj_off[n+1], adj sorted unique per node).  chunk_nodes = 0: one stream, one window (the
// single-threaded store that produced the cnr-2000 fixture).  Outputs are malloc'ed: bvgt_free.
int bvgt_store(const bvg_params* p, int64_t n, const uint64_t* adj_off, const int64_t* adj, int64_t chunk_nodes, int nthreads,
               uint8_t** graph, uint64_t* graph_bytes, uint64_t** offsets, bvgt_stats* stats) {
    StoreOut o;
    int r = store_generic(*p, n, chunk_nodes, nthreads, [&]() { return std::unique_ptr<ListSource>(new AdjSource(adj_off, adj)); }, o);
    if (r) return r;
    *graph = o.graph; *graph_bytes = o.graph_bytes; *offsets = o.offsets; fill_stats(o, stats);
    return 0;
}

// Generates the synthetic web-like graph chunk by chunk and stores it directly (no adjacency in memory).
int bvgt_synth_store(const bvg_params* p, const bvgt_synth_params* sp, int64_t n, uint64_t seed, int64_t chunk_nodes, int nthreads,
                     uint8_t** graph, uint64_t* graph_bytes, uint64_t** offsets, bvgt_stats* stats) {
    if (chunk_nodes <= 0) chunk_nodes = n;
    SynthParams s{sp->p_empty, sp->mean_deg, sp->tail_alpha, sp->max_deg, sp->p_copy, sp->keep_run, sp->skip_run,
                  sp->p_interval, sp->interval_len, sp->local_gap, sp->p_far, sp->window, sp->extra_mean};
    StoreOut o;
    int r = store_generic(*p, n, chunk_nodes, nthreads, [&]() { return std::unique_ptr<ListSource>(new SynthSource(s, n, seed, chunk_nodes)); }, o);
    if (r) return r;
    *graph = o.graph; *graph_bytes = o.graph_bytes; *offsets = o.offsets; fill_stats(o, stats);
    return 0;
}

// Same generator, adjacency out (for tests that want the expected lists): adj_off[n+1] + adj, malloc'ed.
int bvgt_synth_adjacency(const bvgt_synth_params* sp, int64_t n, uint64_t seed, int64_t chunk_nodes, uint64_t** adj_off, int64_t** adj) {
    if (chunk_nodes <= 0) chunk_nodes = n;
    SynthParams s{sp->p_empty, sp->mean_deg, sp->tail_alpha, sp->max_deg, sp->p_copy, sp->keep_run, sp->skip_run,
                  sp->p_interval, sp->interval_len, sp->local_gap, sp->p_far, sp->window, sp->extra_mean};
    SynthSource src(s, n, seed, chunk_nodes);
    std::vector<int64_t> all, cur; std::vector<uint64_t> off((size_t)n + 1);
    for (int64_t x = 0; x < n; x++) {
        if (x % chunk_nodes == 0) src.begin_chunk(x);
        src.next(x, cur);
        off[(size_t)x] = all.size(); all.insert(all.end(), cur.begin(), cur.end());
    }
    off[(size_t)n] = all.size();
    *adj_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    *adj = (int64_t*)malloc(sizeof(int64_t) * (all.size() + 1));
    if (!*adj_off || !*adj) return BVG_E_NOMEM;
    memcpy(*adj_off, off.data(), sizeof(uint64_t) * ((size_t)n + 1));
    memcpy(*adj, all.data(), sizeof(int64_t) * all.size());
    return 0;
}

// writeOffsets, BVGraph.java:2595-2609 / :2228,:2311: n+1 coded gaps, first = offsets[0] = 0.
int bvgt_encode_offsets(const uint64_t* offsets, int64_t n, int coding, uint8_t** bytes, uint64_t* nbytes, uint64_t* nbits) {
    BitWriter w; uint64_t prev = 0;
    for (int64_t i = 0; i <= n; i++) { write_coded(w, offsets[i] - prev, coding, 0); prev = offsets[i]; }
    if (nbits) *nbits = w.nbits;
    w.flush();
    *bytes = (uint8_t*)malloc(w.bytes.size() + 16);
    if (!*bytes) return BVG_E_NOMEM;
    memcpy(*bytes, w.bytes.data(), w.bytes.size()); memset(*bytes + w.bytes.size(), 0, 16);
    *nbytes = w.bytes.size();
    return 0;
}

// Writes `count` values with one coding back to back (unit tests of the decoders' code tables).
int bvgt_encode_values(const uint64_t* vals, int64_t count, int coding, int k, uint8_t** bytes, uint64_t* nbytes) {
    BitWriter w;
    for (int64_t i = 0; i < count; i++) write_coded(w, vals[i], coding, k);
    w.flush();
    *bytes = (uint8_t*)malloc(w.bytes.size() + 16);
    if (!*bytes) return BVG_E_NOMEM;
    memcpy(*bytes, w.bytes.data(), w.bytes.size()); memset(*bytes + w.bytes.size(), 0, 16);
    *nbytes = w.bytes.size();
    return 0;
}

// BitStreamArcLabelledImmutableGraph.store (labelling/BitStreamArcLabelledImmutableGraph.java:655-680): the labels of every arc
// in successor order (label.toBitStream: GammaCodedIntLabel.java:74-76 writeGamma, FixedWidthIntLabel.java:76-78
// writeInt(value, width)); loffsets[n+1] = bit position of each node's run.  kind: 1 gamma, 2 fixed width.
int bvgt_store_labels(int kind, int width, const int32_t* values, const uint64_t* arc_off, int64_t n, uint8_t** bytes, uint64_t* nbytes, uint64_t** loffsets) {
    BitWriter w;
    *loffsets = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n + 1));
    if (!*loffsets) return BVG_E_NOMEM;
    for (int64_t x = 0; x < n; x++) {
        (*loffsets)[x] = w.nbits;
        for (uint64_t a = arc_off[x]; a < arc_off[x + 1]; a++) {
            if (kind == 1) { if (values[a] < 0) { free(*loffsets); return BVG_E_ARG; } write_gamma(w, (uint64_t)values[a]); }
            else if (width > 0) w.put((uint64_t)(uint32_t)values[a] & (width == 32 ? 0xFFFFFFFFull : ((1ULL << width) - 1)), width);
        }
    }
    (*loffsets)[n] = w.nbits;
    w.flush();
    *bytes = (uint8_t*)malloc(w.bytes.size() + 16);
    if (!*bytes) { free(*loffsets); return BVG_E_NOMEM; }
    memcpy(*bytes, w.bytes.data(), w.bytes.size()); memset(*bytes + w.bytes.size(), 0, 16);
    *nbytes = w.bytes.size();
    return 0;
}

// List labels (FixedWidthIntListLabel.toBitStream, FixedWidthIntListLabel.java:81-85): per arc gamma(length) + the elements.
// list_off[m+1] = exclusive prefix of the list lengths over the arcs.
int bvgt_store_label_lists(int width, const uint64_t* list_off, const int32_t* values, const uint64_t* arc_off, int64_t n, uint8_t** bytes, uint64_t* nbytes, uint64_t** loffsets) {
    BitWriter w;
    *loffsets = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n + 1));
    if (!*loffsets) return BVG_E_NOMEM;
    for (int64_t x = 0; x < n; x++) {
        (*loffsets)[x] = w.nbits;
        for (uint64_t a = arc_off[x]; a < arc_off[x + 1]; a++) {
            write_gamma(w, list_off[a + 1] - list_off[a]);
            if (width > 0) for (uint64_t t = list_off[a]; t < list_off[a + 1]; t++) w.put((uint64_t)(uint32_t)values[t] & (width == 32 ? 0xFFFFFFFFull : ((1ULL << width) - 1)), width);
        }
    }
    (*loffsets)[n] = w.nbits;
    w.flush();
    *bytes = (uint8_t*)malloc(w.bytes.size() + 16);
    if (!*bytes) { free(*loffsets); return BVG_E_NOMEM; }
    memcpy(*bytes, w.bytes.data(), w.bytes.size()); memset(*bytes + w.bytes.size(), 0, 16);
    *nbytes = w.bytes.size();
    return 0;
}

// FixedWidthLongListLabel.toBitStream (labelling/FixedWidthLongListLabel.java:90-95): per arc gamma(length) + writeLong(value, width)
int bvgt_store_label_lists64(int width, const uint64_t* list_off, const int64_t* values, const uint64_t* arc_off, int64_t n, uint8_t** bytes, uint64_t* nbytes, uint64_t** loffsets) {
    if (width < 0 || width > 64) return BVG_E_ARG;
    BitWriter w;
    *loffsets = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n + 1));
    if (!*loffsets) return BVG_E_NOMEM;
    const uint64_t mask = width == 64 ? ~0ULL : ((1ULL << width) - 1);
    for (int64_t x = 0; x < n; x++) {
        (*loffsets)[x] = w.nbits;
        for (uint64_t a = arc_off[x]; a < arc_off[x + 1]; a++) {
            write_gamma(w, list_off[a + 1] - list_off[a]);
            if (width > 0) for (uint64_t t = list_off[a]; t < list_off[a + 1]; t++) {
                const uint64_t v = (uint64_t)values[t] & mask;
                if (width > 32) { w.put(v >> 32, width - 32); w.put(v & 0xFFFFFFFFull, 32); } else w.put(v, width);
            }
        }
    }
    (*loffsets)[n] = w.nbits;
    w.flush();
    *bytes = (uint8_t*)malloc(w.bytes.size() + 16);
    if (!*bytes) { free(*loffsets); return BVG_E_NOMEM; }
    memcpy(*bytes, w.bytes.data(), w.bytes.size()); memset(*bytes + w.bytes.size(), 0, 16);
    *nbytes = w.bytes.size();
    return 0;
}

void bvgt_free(void* p) { free(p); }

}  // extern "C"
