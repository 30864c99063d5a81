// bvgraph.hpp — C++ host-side mirror of the reference's graph API for the decode path, over the C ABI
// of libbvgraph_hip.so (include/bvgraph_hip.h).  Header-only; link with -lbvgraph_hip.
//
// The reference is Java; there is no JVM in this image, so the host side above the C ABI is written in
// C++ with the reference's names, argument meaning and error behaviour (paths relative to
// /root/reference/src/it/unimi/dsi/big/webgraph):
//   LazyLongIterator.java:28-44          nextLong() -> next successor or -1; skip(n)
//   NodeIterator.java:34-133             hasNext / nextLong / outdegree / successors / successorBigArray / copy(upperBound) / skip
//   ImmutableGraph.java:245-447          numNodes / numArcs / randomAccess / outdegree / successors / successorBigArray /
//                                        nodeIterator(from) / splitNodeIterators(k) / copy()
//   BVGraph.java:1345-1464               load / loadMapped / loadOffline / loadSequential
// Error mapping: BVG_E_ARG -> std::invalid_argument (IllegalArgumentException),
// BVG_E_STATE -> std::logic_error (IllegalStateException), BVG_E_UNSUPPORTED -> UnsupportedOperation,
// BVG_E_IO/EOF -> std::ios_base::failure (IOException), nextLong() past the end -> std::out_of_range
// (NoSuchElementException).  A JNI shim is the same calls with jlong/jlongArray marshalling.
#pragma once
#include <cstdint>
#include <ios>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bvgraph_hip.h"

namespace webgraph {

struct UnsupportedOperation : std::runtime_error { using std::runtime_error::runtime_error; };
struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

inline void check(int st, const char* what) {
    if (st == BVG_OK) return;
    std::string msg = std::string(what) + ": " + bvg_strerror(st);
    switch (st) {
        case BVG_E_ARG: throw std::invalid_argument(msg);
        case BVG_E_STATE: throw std::logic_error(msg);
        case BVG_E_UNSUPPORTED: throw UnsupportedOperation(msg);
        case BVG_E_IO: case BVG_E_EOF: throw std::ios_base::failure(msg);
        case BVG_E_NOMEM: throw std::bad_alloc();
        default: throw DeviceError(msg);
    }
}

// LazyLongIterators.wrap(array, n), LazyLongIterators.java:220-255
class LazyLongIterator {
    const int64_t* a_; int64_t n_, i_ = 0;
public:
    LazyLongIterator(const int64_t* a, int64_t n) : a_(a), n_(n) {}
    int64_t nextLong() { return i_ < n_ ? a_[i_++] : -1; }
    int64_t skip(int64_t n) { int64_t k = n < n_ - i_ ? n : n_ - i_; i_ += k; return k; }
};

class BVGraph;

// BVGraph.BVGraphNodeIterator (BVGraph.java:1100-1245): sequential scan served by batched GPU decodes.
class NodeIterator {
    std::shared_ptr<BVGraph> g_;
    int64_t from_, curr_, limit_, b0_ = 0, b1_ = 0, batch_;
    std::vector<int32_t> deg_; std::vector<int64_t> succ_; std::vector<uint64_t> cum_;
    void fill(int64_t x);
public:
    NodeIterator(std::shared_ptr<BVGraph> g, int64_t from, int64_t upperBound, int64_t batchNodes = 1 << 16);
    bool hasNext() const { return curr_ < limit_; }                                        // BVGraph.java:1179-1181
    int64_t nextLong() {                                                                   // BVGraph.java:1164-1176
        if (!hasNext()) throw std::out_of_range("NoSuchElementException");
        ++curr_;
        if (curr_ < b0_ || curr_ >= b1_) fill(curr_);
        return curr_;
    }
    int64_t outdegree() const { started(); return deg_[(size_t)(curr_ - b0_)]; }          // BVGraph.java:1206-1209
    // successorBigArray(): valid until the next nextLong() (NodeIterator.java:80-96)
    const int64_t* successorBigArray() const { started(); return succ_.data() + cum_[(size_t)(curr_ - b0_)]; }
    LazyLongIterator successors() const { return LazyLongIterator(successorBigArray(), outdegree()); }
    NodeIterator copy(int64_t upperBound) const;                                           // BVGraph.java:1223-1229
    int64_t skip(int64_t n) { int64_t k = 0; while (k < n && hasNext()) { nextLong(); k++; } return k; }
private:
    void started() const { if (curr_ == from_ - 1) throw std::logic_error("IllegalStateException"); }   // BVGraph.java:1185
};

class BVGraph : public std::enable_shared_from_this<BVGraph> {
    bvg_graph* h_ = nullptr; bvg_params p_{}; std::string basename_;
    explicit BVGraph(bvg_graph* h) : h_(h) { check(bvg_info(h_, &p_), "info"); }
public:
    ~BVGraph() { bvg_close(h_); }
    BVGraph(const BVGraph&) = delete;
    static std::shared_ptr<BVGraph> load(const std::string& basename, int device = 0, int mode = BVG_LOAD_STANDARD) {   // BVGraph.java:1345
        bvg_graph* h = nullptr; check(bvg_open(basename.c_str(), mode, device, &h), "load");
        auto g = std::shared_ptr<BVGraph>(new BVGraph(h)); g->basename_ = basename; return g;
    }
    static std::shared_ptr<BVGraph> loadMapped(const std::string& b, int device = 0) { return load(b, device, BVG_LOAD_MAPPED); }
    static std::shared_ptr<BVGraph> loadOffline(const std::string& b, int device = 0) { return load(b, device, BVG_LOAD_OFFLINE); }
    static std::shared_ptr<BVGraph> loadSequential(const std::string& b, int device = 0) { return load(b, device, BVG_LOAD_SEQUENTIAL); }
    static std::shared_ptr<BVGraph> fromMemory(const bvg_params& p, const uint8_t* graph, uint64_t nbytes, const uint64_t* offsets, int device = 0) {
        bvg_graph* h = nullptr; check(bvg_open_mem(&p, graph, nbytes, offsets, device, &h), "open_mem");
        return std::shared_ptr<BVGraph>(new BVGraph(h));
    }
    bvg_graph* handle() const { return h_; }
    int64_t numNodes() const { return p_.nodes; }
    int64_t numArcs() const { if (p_.arcs < 0) throw UnsupportedOperation("numArcs"); return p_.arcs; }     // ImmutableGraph.java:253-258
    bool randomAccess() const { return true; }
    bool hasCopiableIterators() const { return true; }
    const std::string& basename() const { return basename_; }
    int windowSize() const { return p_.window_size; }
    int maxRefCount() const { return p_.max_ref_count; }
    int minIntervalLength() const { return p_.min_interval_length; }
    std::shared_ptr<BVGraph> copy() const {                                                // BVGraph.java:553-578
        bvg_graph* h = nullptr; check(bvg_copy(h_, &h), "copy");
        auto g = std::shared_ptr<BVGraph>(new BVGraph(h)); g->basename_ = basename_; return g;
    }
    int64_t outdegree(int64_t x) {                                                         // BVGraph.java:821-842
        if (x < 0 || x >= p_.nodes) throw std::invalid_argument("Node index out of range");
        int32_t d; check(bvg_outdegrees(h_, x, x + 1, &d), "outdegree"); return d;
    }
    const bvg_params& params() const { return p_; }
    // how much index the scans of this handle build and use (bvg_tuning.no_index): 0 = the full residual skip index, 1 = none, 2 = marks only (validation marks + entries for
    // lists of >= 4 096 residuals: ~0.03 % of the stream instead of ~50 %)
    void setIndexMode(int mode) { bvg_tuning t{}; t.no_index = (uint32_t)mode; check(bvg_set_tuning(h_, &t), "set_tuning"); }
    // BVGraph.store on the device (bvg_store; BVG:2404-2457 with chunkNodes > 0, the single-threaded store with 0): the bytes of
    // basename.graph and the bit offsets (basename.offsets holds their gamma-coded gaps)
    static void store(const bvg_params& p, const std::vector<uint64_t>& adjOff, const std::vector<int64_t>& adj, std::vector<uint8_t>& graph,
                      std::vector<uint64_t>& offsets, int64_t chunkNodes = 0, int device = 0) {
        uint8_t* g = nullptr; uint64_t nb = 0; uint64_t* o = nullptr;
        const int64_t n = (int64_t)adjOff.size() - 1;
        static const int64_t none = 0;
        check(bvg_store(&p, n, adjOff.data(), adj.empty() ? &none : adj.data(), chunkNodes, device, &g, &nb, &o), "store");
        graph.assign(g, g + nb); offsets.assign(o, o + n + 1);
        bvg_free(g); bvg_free(o);
    }
    // decode of [from,to): outdegrees + concatenated successor lists
    void decodeRange(int64_t from, int64_t to, std::vector<int32_t>& deg, std::vector<int64_t>& succ) {
        deg.resize((size_t)(to > from ? to - from : 0));
        uint64_t need = 0;
        int st = bvg_decode_range(h_, from, to, deg.data(), nullptr, 0, &need);
        if (st != BVG_E_CAPACITY) check(st, "decode_range");
        succ.resize((size_t)need);
        if (need) check(bvg_decode_range(h_, from, to, deg.data(), succ.data(), need, &need), "decode_range");
    }
    std::vector<int64_t> successorBigArray(int64_t x) {                                    // BVGraph.java:860-867
        if (x < 0 || x >= p_.nodes) throw std::invalid_argument("Node index out of range");
        std::vector<int32_t> d; std::vector<int64_t> s; decodeRange(x, x + 1, d, s); return s;
    }
    // successors(x) for a whole frontier (bvg_successors_batch): outdegrees + concatenated lists in request order
    void successorsBatch(const std::vector<int64_t>& nodes, std::vector<int32_t>& deg, std::vector<int64_t>& succ) {
        deg.resize(nodes.size());
        uint64_t need = 0;
        int st = bvg_successors_batch(h_, nodes.data(), (int64_t)nodes.size(), deg.data(), nullptr, 0, &need);
        if (st != BVG_E_CAPACITY) check(st, "successors_batch");
        succ.resize((size_t)need);
        if (need) check(bvg_successors_batch(h_, nodes.data(), (int64_t)nodes.size(), deg.data(), succ.data(), need, &need), "successors_batch");
    }
    // the transpose in CSR form (decode + device sort; Transform.transposeOffline, Transform.java:1058-1160)
    void transposeCSR(std::vector<uint64_t>& toffsets, std::vector<int64_t>& tsucc) {
        toffsets.resize((size_t)p_.nodes + 1);
        uint64_t need = 0;
        int st = bvg_transpose(h_, toffsets.data(), nullptr, 0, &need);
        if (st != BVG_E_CAPACITY) check(st, "transpose");
        tsucc.resize((size_t)need);
        if (need) check(bvg_transpose(h_, toffsets.data(), tsucc.data(), need, &need), "transpose");
    }
    // the symmetrised graph in CSR form (Transform.symmetrizeOffline, Transform.java:546-575: union with the transpose)
    void symmetrizeCSR(std::vector<uint64_t>& soffsets, std::vector<int64_t>& ssucc) {
        soffsets.resize((size_t)p_.nodes + 1);
        uint64_t need = 0;
        int st = bvg_symmetrize(h_, soffsets.data(), nullptr, 0, &need);
        if (st != BVG_E_CAPACITY) check(st, "symmetrize");
        ssucc.resize((size_t)need);
        if (need) check(bvg_symmetrize(h_, soffsets.data(), ssucc.data(), need, &need), "symmetrize");
    }
    NodeIterator nodeIterator(int64_t from = 0) { return NodeIterator(shared_from_this(), from, INT64_MAX); }   // BVGraph.java:1257
    std::vector<NodeIterator> splitNodeIterators(int howMany) {                            // ImmutableGraph.java:405-436
        std::vector<NodeIterator> v; const int64_t n = p_.nodes, m = (n + howMany - 1) / howMany;
        for (int i = 0; i < howMany; i++) {
            int64_t lo = (int64_t)i * m < n ? (int64_t)i * m : n, hi = lo + m < n ? lo + m : n;
            v.emplace_back(lo < n ? copy() : shared_from_this(), lo, hi);
        }
        return v;
    }
    bvg_scan_result scan(int64_t from = 0, int64_t to = -1) {                               // the SpeedTest loop, test/SpeedTest.java:127-141
        bvg_scan_result r; check(bvg_scan(h_, from, to < 0 ? p_.nodes : to, &r), "scan"); return r;
    }
};

inline NodeIterator::NodeIterator(std::shared_ptr<BVGraph> g, int64_t from, int64_t upperBound, int64_t batchNodes)
    : g_(std::move(g)), from_(from), curr_(from - 1), batch_(batchNodes) {
    const int64_t n = g_->numNodes();
    if (from < 0 || from > n) throw std::invalid_argument("Node index out of range");     // BVGraph.java:1128
    limit_ = (upperBound < n ? upperBound : n) - 1;                                        // BVGraph.java:1148
    b0_ = b1_ = from;
}
inline void NodeIterator::fill(int64_t x) {
    int64_t hi = x + batch_ < limit_ + 1 ? x + batch_ : limit_ + 1;
    g_->decodeRange(x, hi, deg_, succ_);
    cum_.assign(deg_.size() + 1, 0);
    for (size_t i = 0; i < deg_.size(); i++) cum_[i + 1] = cum_[i] + (uint64_t)deg_[i];
    b0_ = x; b1_ = hi;
}
inline NodeIterator NodeIterator::copy(int64_t upperBound) const { return NodeIterator(g_->copy(), curr_ + 1, upperBound, batch_); }

// labelling/BitStreamArcLabelledImmutableGraph.java: an underlying BVGraph plus one int label per arc (GammaCodedIntLabel /
// FixedWidthIntLabel) and the list labels (FixedWidthIntListLabel, FixedWidthLongListLabel), all decoded on the device.  decodeRange is one batch of the labelled node iterator (:565-582);
// successors(x) positions at the node's label offset (:208-229).
class BitStreamArcLabelledImmutableGraph {
    std::shared_ptr<BVGraph> g_; bvg_labels* l_ = nullptr;
    BitStreamArcLabelledImmutableGraph(std::shared_ptr<BVGraph> g, bvg_labels* l) : g_(std::move(g)), l_(l) {}
public:
    ~BitStreamArcLabelledImmutableGraph() { bvg_labels_close(l_); }
    BitStreamArcLabelledImmutableGraph(const BitStreamArcLabelledImmutableGraph&) = delete;
    static std::shared_ptr<BitStreamArcLabelledImmutableGraph> load(const std::string& basename, int device = 0) {     // :378-484
        char under[4096];                                                                      // the property file names the underlying graph
        check(bvg_labels_read_properties(basename.c_str(), nullptr, nullptr, under, sizeof under), "labels properties");
        auto g = BVGraph::load(under, device);
        bvg_labels* l = nullptr; check(bvg_labels_open(basename.c_str(), g->numNodes(), device, &l, nullptr, 0), "labels");
        return std::shared_ptr<BitStreamArcLabelledImmutableGraph>(new BitStreamArcLabelledImmutableGraph(g, l));
    }
    std::shared_ptr<BVGraph> underlying() const { return g_; }
    int64_t numNodes() const { return g_->numNodes(); }
    void decodeRange(int64_t from, int64_t to, std::vector<int32_t>& deg, std::vector<int64_t>& succ, std::vector<int32_t>& lab) {
        g_->decodeRange(from, to, deg, succ);
        lab.resize(succ.size());
        uint64_t n = 0;
        check(bvg_labels_decode_range(l_, from, to, deg.data(), lab.data(), lab.size(), &n), "labels_decode_range");
    }
    // list labels (FixedWidthIntListLabel.java:73-78, FixedWidthLongListLabel.java:81-87): listOff[arcs + 1] = where each arc's list
    // starts in `values`
    void decodeRangeLists(int64_t from, int64_t to, std::vector<int32_t>& deg, std::vector<int64_t>& succ, std::vector<uint64_t>& listOff, std::vector<int64_t>& values) {
        g_->decodeRange(from, to, deg, succ);
        listOff.assign(succ.size() + 1, 0);
        int kind = 0, width = 0; int64_t nodes = 0; uint64_t sb = 0;
        check(bvg_labels_info(l_, &kind, &width, &nodes, &sb), "labels_info");
        uint64_t n = 0;
        if (kind == BVG_LABEL_FIXED_LONG_LIST) {
            int st = bvg_labels_decode_range_lists64(l_, from, to, deg.data(), listOff.data(), nullptr, 0, &n);
            if (st != BVG_E_CAPACITY) check(st, "labels_decode_range_lists64");
            values.resize(n);
            if (n) check(bvg_labels_decode_range_lists64(l_, from, to, deg.data(), listOff.data(), values.data(), n, &n), "labels_decode_range_lists64");
        } else {
            int st = bvg_labels_decode_range_lists(l_, from, to, deg.data(), listOff.data(), nullptr, 0, &n);
            if (st != BVG_E_CAPACITY) check(st, "labels_decode_range_lists");
            std::vector<int32_t> v32(n);
            if (n) check(bvg_labels_decode_range_lists(l_, from, to, deg.data(), listOff.data(), v32.data(), n, &n), "labels_decode_range_lists");
            values.assign(v32.begin(), v32.end());
        }
    }
};

}  // namespace webgraph
