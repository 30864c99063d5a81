// bvg_labels.hip — arc labels stored as a bit stream, decoded beside the successors (first half).
//
// Reference: labelling/BitStreamArcLabelledImmutableGraph.java (paths relative to
// /root/reference/src/it/unimi/dsi/big/webgraph): the label file holds, node after node, the labels of the node's arcs in
// successor order (:75-84); basename.labeloffsets holds the gamma-coded bit lengths of those per-node runs after a leading
// gamma(0) (store(), :655-680), i.e. the same shape as BVGraph's .offsets.  The node iterator reads `outdegree` labels at every
// nextLong() (:565-582, label.fromBitStream), the random-access iterator positions the stream at offset[x] (:208-229).
// Label classes handled on the device: GammaCodedIntLabel (one gamma-coded natural per arc, GammaCodedIntLabel.java:60-64) and
// FixedWidthIntLabel (readInt(width), FixedWidthIntLabel.java:70-73), and the list label FixedWidthIntListLabel (gamma length +
// elements of `width` bits, FixedWidthIntListLabel.java:73-78) and FixedWidthLongListLabel (the same with readLong(width), width <= 64,
// FixedWidthLongListLabel.java:81-87) in two passes (lengths, prefix sum, elements).
//
// Layout in HBM: the label stream verbatim (zero-padded to 16 bytes + 16), uint64 label_offsets[n+1].  One thread decodes the
// labels of one node through the generic BitCursor of bvg_device.h and writes them at the node's arc offset (exclusive
// prefix of the outdegrees, computed on the device); every node checks that its run ends where the next offset says.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "bvg_kernels.h"

struct bvg_labels {
    int device = 0, kind = 0, width = 0;
    int64_t nodes = 0;
    uint64_t nbytes = 0, padded = 0;
    uint8_t* d_stream = nullptr;
    uint64_t* d_offsets = nullptr;
    hipStream_t stream = nullptr;
    unsigned* d_err = nullptr;
    // workspace (grown on demand)
    uint64_t* d_cum = nullptr; uint64_t* d_tmp = nullptr;
    size_t deg_cap = 0, tmp_cap = 0;
};

namespace bvg {
namespace {

#define LCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { if (dbg_on()) fprintf(stderr, "[bvg] %s -> %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
                        return _e == hipErrorOutOfMemory ? BVG_E_NOMEM : BVG_E_HIP; } } while (0)

__global__ void __launch_bounds__(256) labels_kernel(const uint8_t* stream, uint64_t limit_byte, const uint64_t* loff, int64_t from, int64_t count,
                                                     const int32_t* deg, const uint64_t* cum, int kind, int width, int32_t* out, unsigned* err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t x = from + i;
    const uint64_t end = loff[x + 1];
    BitCursor c{stream, loff[x], limit_byte};
    const uint32_t d = (uint32_t)deg[i];
    int32_t* const o = out + cum[i];
    unsigned e = 0;
    for (uint32_t j = 0; j < d; j++) {                                        // BitStreamArcLabelledImmutableGraph.java:579
        uint64_t v;
        if (kind == BVG_LABEL_GAMMA_INT) { v = c.read_gamma(end); if (v > 0x7FFFFFFFull) e |= ERR_MALFORMED; }   // readGamma() is an int
        else v = c.read_bits((unsigned)width);
        o[j] = (int32_t)(uint32_t)v;
        if (c.pos > end) { e |= ERR_OVERRUN; break; }
    }
    if (c.pos != end) e |= ERR_MALFORMED;                                     // the run must end where the next one starts
    if (e) atomicOr(err, e);
}

// FixedWidthIntListLabel (FixedWidthIntListLabel.java:73-78): per arc gamma(length) then `length` elements of `width` bits.
// PASS 0 writes the length of every arc's list (lens[arc]); PASS 1 writes the elements at vals + voff[arc].
// V = int32_t: FixedWidthIntListLabel (width <= 32); V = int64_t: FixedWidthLongListLabel (readLong(width), width <= 64,
// FixedWidthLongListLabel.java:81-87).
template <int PASS, typename V>
__global__ void __launch_bounds__(256) label_lists_kernel(const uint8_t* stream, uint64_t limit_byte, const uint64_t* loff, int64_t from, int64_t count,
                                                          const int32_t* deg, const uint64_t* cum, int width, int32_t* lens, const uint64_t* voff, V* vals, unsigned* err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t x = from + i;
    const uint64_t end = loff[x + 1];
    BitCursor c{stream, loff[x], limit_byte};
    const uint32_t d = (uint32_t)deg[i];
    const uint64_t a0 = cum[i];
    unsigned e = 0;
    for (uint32_t j = 0; j < d; j++) {
        const uint64_t len = c.read_gamma(end);
        if (len > 0x7FFFFFFFull || c.pos + len * (uint64_t)width > end) { e |= ERR_OVERRUN; if (PASS == 0) for (uint32_t t = j; t < d; t++) lens[a0 + t] = 0; break; }
        if (PASS == 0) { lens[a0 + j] = (int32_t)len; c.pos += len * (uint64_t)width; }
        else { V* const o = vals + voff[a0 + j]; for (uint64_t t = 0; t < len; t++) o[t] = sizeof(V) == 4 ? (V)(int32_t)(uint32_t)c.read_bits((unsigned)width) : (V)c.read_bits((unsigned)width); }
    }
    if (!e && c.pos != end) e |= ERR_MALFORMED;
    if (e) atomicOr(err, e);
}

int read_all(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return BVG_E_IO;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = n > 0 ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == out.size() ? 0 : BVG_E_IO;
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) a++;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) b--;
    return s.substr(a, b - a);
}

}  // namespace
}  // namespace bvg

using namespace bvg;

template <typename V> int labels_lists_impl(bvg_labels* l, int want_kind, int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, V* values, uint64_t cap, uint64_t* n_values) {
    if (!l || from < 0 || to < from || to > l->nodes || (to > from && !outdeg) || !list_off) return BVG_E_ARG;
    if (l->kind != want_kind) return BVG_E_UNSUPPORTED;
    LCHK(hipSetDevice(l->device));
    const int64_t cnt = to - from;
    if (n_values) *n_values = 0;
    list_off[0] = 0;
    if (cnt == 0) return 0;
    uint64_t arcs = 0;
    for (int64_t i = 0; i < cnt; i++) { if (outdeg[i] < 0) return BVG_E_ARG; arcs += (uint64_t)outdeg[i]; }
    if (arcs == 0) return 0;
    int32_t* d_deg = nullptr; int32_t* d_lens = nullptr; V* d_vals = nullptr; uint64_t* d_cum = nullptr; uint64_t* d_voff = nullptr; uint64_t* d_tmp = nullptr;
    auto done = [&](int code) { for (void* p : {(void*)d_deg, (void*)d_lens, (void*)d_vals, (void*)d_cum, (void*)d_voff, (void*)d_tmp}) if (p) (void)hipFree(p); return code; };
    const size_t tneed = std::max(scan_tmp_elems(cnt), scan_tmp_elems((int64_t)arcs));
    if (hipMalloc(&d_deg, (size_t)cnt * 4) != hipSuccess || hipMalloc(&d_cum, (size_t)(cnt + 1) * 8) != hipSuccess || hipMalloc(&d_lens, (size_t)arcs * 4) != hipSuccess ||
        hipMalloc(&d_voff, (size_t)(arcs + 1) * 8) != hipSuccess || hipMalloc(&d_tmp, tneed * 8) != hipSuccess) return done(BVG_E_NOMEM);
    if (hipMemcpy(d_deg, outdeg, (size_t)cnt * 4, hipMemcpyHostToDevice) != hipSuccess) return done(BVG_E_HIP);
    launch_exclusive_scan(d_deg, d_cum, cnt, d_tmp, l->stream);
    if (hipMemsetAsync(l->d_err, 0, sizeof(unsigned), l->stream) != hipSuccess) return done(BVG_E_HIP);
    const uint64_t limit = l->padded - 16;
    const dim3 grid((unsigned)((cnt + 255) / 256));
    hipLaunchKernelGGL((label_lists_kernel<0, V>), grid, dim3(256), 0, l->stream, l->d_stream, limit, l->d_offsets, from, cnt, d_deg, d_cum, l->width, d_lens, (const uint64_t*)nullptr, (V*)nullptr, l->d_err);
    launch_exclusive_scan(d_lens, d_voff, (int64_t)arcs, d_tmp, l->stream);
    unsigned herr = 0;
    if (hipMemcpyAsync(&herr, l->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, l->stream) != hipSuccess) return done(BVG_E_HIP);
    if (hipMemcpyAsync(list_off, d_voff, (size_t)(arcs + 1) * 8, hipMemcpyDeviceToHost, l->stream) != hipSuccess) return done(BVG_E_HIP);
    if (hipStreamSynchronize(l->stream) != hipSuccess) return done(BVG_E_HIP);
    if (herr) return done(BVG_E_EOF);
    const uint64_t total = list_off[arcs];
    if (n_values) *n_values = total;
    if (total > cap || (total && !values)) return done(BVG_E_CAPACITY);
    if (total == 0) return done(0);
    if (hipMalloc(&d_vals, (size_t)total * sizeof(V)) != hipSuccess) return done(BVG_E_NOMEM);
    hipLaunchKernelGGL((label_lists_kernel<1, V>), grid, dim3(256), 0, l->stream, l->d_stream, limit, l->d_offsets, from, cnt, d_deg, d_cum, l->width, (int32_t*)nullptr, d_voff, d_vals, l->d_err);
    if (hipMemcpyAsync(values, d_vals, (size_t)total * sizeof(V), hipMemcpyDeviceToHost, l->stream) != hipSuccess) return done(BVG_E_HIP);
    if (hipMemcpyAsync(&herr, l->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, l->stream) != hipSuccess) return done(BVG_E_HIP);
    if (hipStreamSynchronize(l->stream) != hipSuccess) return done(BVG_E_HIP);
    return done(herr ? BVG_E_EOF : 0);
}

extern "C" {

// "it.unimi.dsi.big.webgraph.labelling.GammaCodedIntLabel(FOO)" / "...FixedWidthIntLabel(FOO,10)" (Label.toSpec; the class may be
// given with either the big or the standard package, BitStreamArcLabelledImmutableGraph.java:115-118)
int bvg_labels_parse_spec(const char* spec, int* kind, int* width) {
    if (!spec || !kind || !width) return BVG_E_ARG;
    const std::string s = trim(spec);
    const size_t lp = s.find('('), rp = s.rfind(')');
    if (lp == std::string::npos || rp == std::string::npos || rp < lp) return BVG_E_IO;
    std::string cls = trim(s.substr(0, lp));
    const size_t dot = cls.rfind('.');
    if (dot != std::string::npos) cls = cls.substr(dot + 1);
    const std::string args = s.substr(lp + 1, rp - lp - 1);
    if (cls == "GammaCodedIntLabel") { *kind = BVG_LABEL_GAMMA_INT; *width = 0; return 0; }
    if (cls == "FixedWidthIntLabel") {
        const size_t comma = args.find(',');
        if (comma == std::string::npos) return BVG_E_IO;
        const std::string w = trim(args.substr(comma + 1));
        char* endp = nullptr; const long v = strtol(w.c_str(), &endp, 10);
        if (endp == w.c_str() || v < 0 || v > 32) return BVG_E_IO;            // FixedWidthIntLabel.java:47 (width in [0..32])
        *kind = BVG_LABEL_FIXED_INT; *width = (int)v; return 0;
    }
    if (cls == "FixedWidthIntListLabel") {
        const size_t comma = args.find(',');
        if (comma == std::string::npos) return BVG_E_IO;
        const std::string w = trim(args.substr(comma + 1));
        char* endp = nullptr; const long v = strtol(w.c_str(), &endp, 10);
        if (endp == w.c_str() || v < 0 || v > 32) return BVG_E_IO;
        *kind = BVG_LABEL_FIXED_INT_LIST; *width = (int)v; return 0;
    }
    if (cls == "FixedWidthLongListLabel") {
        const size_t comma = args.find(',');
        if (comma == std::string::npos) return BVG_E_IO;
        const std::string w = trim(args.substr(comma + 1));
        char* endp = nullptr; const long v = strtol(w.c_str(), &endp, 10);
        if (endp == w.c_str() || v < 0 || v > 64) return BVG_E_IO;            // FixedWidthLongListLabel.java:50 (width in [0..64])
        *kind = BVG_LABEL_FIXED_LONG_LIST; *width = (int)v; return 0;
    }
    return BVG_E_UNSUPPORTED;                                                 // user classes
}

int bvg_labels_open_mem(int kind, int width, int64_t nodes, const uint8_t* stream, uint64_t nbytes, const uint64_t* label_offsets, int device, bvg_labels** out) {
    if (!out || nodes < 0 || !label_offsets || (nbytes && !stream)) return BVG_E_ARG;
    if (kind != BVG_LABEL_GAMMA_INT && kind != BVG_LABEL_FIXED_INT && kind != BVG_LABEL_FIXED_INT_LIST && kind != BVG_LABEL_FIXED_LONG_LIST) return BVG_E_UNSUPPORTED;
    if (kind != BVG_LABEL_GAMMA_INT && (width < 0 || width > (kind == BVG_LABEL_FIXED_LONG_LIST ? 64 : 32))) return BVG_E_ARG;
    if (label_offsets[nodes] > nbytes * 8) return BVG_E_EOF;
    for (int64_t i = 0; i < nodes; i++) if (label_offsets[i] > label_offsets[i + 1]) return BVG_E_IO;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return BVG_E_HIP;
    LCHK(hipSetDevice(device));
    bvg_labels* l = new bvg_labels();
    l->device = device; l->kind = kind; l->width = width; l->nodes = nodes; l->nbytes = nbytes;
    l->padded = ((nbytes + 15) & ~15ull) + 16;
    auto fail = [&](int code) { bvg_labels_close(l); return code; };
    if (hipMalloc(&l->d_stream, l->padded) != hipSuccess) return fail(BVG_E_NOMEM);
    if (hipMemset(l->d_stream, 0, l->padded) != hipSuccess) return fail(BVG_E_HIP);
    if (nbytes && hipMemcpy(l->d_stream, stream, nbytes, hipMemcpyHostToDevice) != hipSuccess) return fail(BVG_E_HIP);
    if (hipMalloc(&l->d_offsets, (size_t)(nodes + 1) * sizeof(uint64_t)) != hipSuccess) return fail(BVG_E_NOMEM);
    if (hipMemcpy(l->d_offsets, label_offsets, (size_t)(nodes + 1) * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return fail(BVG_E_HIP);
    if (hipMalloc(&l->d_err, sizeof(unsigned)) != hipSuccess) return fail(BVG_E_NOMEM);
    if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) return fail(BVG_E_HIP);
    *out = l;
    return 0;
}

// basename.properties of a BitStreamArcLabelledImmutableGraph (:95-118): the label class (labelspec) and the basename of the
// underlying graph (underlyinggraph, resolved against the property file).  Host-only.
int bvg_labels_read_properties(const char* basename, int* kind, int* width, char* underlying, size_t underlying_cap) {
    if (!basename) return BVG_E_ARG;
    const std::string base(basename);
    std::vector<uint8_t> props;
    int r = read_all(base + ".properties", props); if (r) return r;
    std::string spec, under;
    {
        const std::string text(props.begin(), props.end());
        size_t p = 0;
        while (p < text.size()) {
            size_t q = text.find('\n', p); if (q == std::string::npos) q = text.size();
            const std::string line = trim(text.substr(p, q - p)); p = q + 1;
            if (line.empty() || line[0] == '#' || line[0] == '!') continue;
            size_t eq = line.find_first_of("=:"); if (eq == std::string::npos) continue;
            const std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
            if (k == "labelspec") spec = v; else if (k == "underlyinggraph") under = v;
        }
    }
    if (spec.empty()) return BVG_E_IO;                                        // :405 "does not contain a label specification"
    int k = 0, w = 0;
    r = bvg_labels_parse_spec(spec.c_str(), &k, &w); if (r) return r;
    if (kind) *kind = k;
    if (width) *width = w;
    if (underlying && underlying_cap) {
        std::string u = under;
        if (!u.empty() && u[0] != '/') { const size_t sl = base.rfind('/'); if (sl != std::string::npos) u = base.substr(0, sl + 1) + u; }   // relative to the property file (:95-97)
        if (u.size() + 1 > underlying_cap) return BVG_E_ARG;
        memcpy(underlying, u.c_str(), u.size() + 1);
    }
    return 0;
}

// BitStreamArcLabelledImmutableGraph.load (:378-484): basename.properties {underlyinggraph, labelspec}, basename.labels,
// basename.labeloffsets.  The underlying graph is opened by the caller (bvg_open) from the returned basename.
int bvg_labels_open(const char* basename, int64_t nodes, int device, bvg_labels** out, char* underlying, size_t underlying_cap) {
    if (!basename || !out || nodes < 0) return BVG_E_ARG;
    const std::string base(basename);
    int kind = 0, width = 0;
    int r = bvg_labels_read_properties(basename, &kind, &width, underlying, underlying_cap); if (r) return r;
    std::vector<uint8_t> lab, offs;
    r = read_all(base + ".labels", lab); if (r) return r;
    r = read_all(base + ".labeloffsets", offs); if (r) return r;
    std::vector<uint64_t> lo((size_t)nodes + 1);
    r = bvg_decode_offsets(offs.data(), offs.size(), nodes, BVG_GAMMA, lo.data()); if (r) return r;     // LabelOffsetsLongIterator (:330-364)
    return bvg_labels_open_mem(kind, width, nodes, lab.data(), lab.size(), lo.data(), device, out);
}

void bvg_labels_close(bvg_labels* l) {
    if (!l) return;
    (void)hipSetDevice(l->device);
    if (l->stream) { (void)hipStreamSynchronize(l->stream); (void)hipStreamDestroy(l->stream); }
    for (void* p : {(void*)l->d_stream, (void*)l->d_offsets, (void*)l->d_err, (void*)l->d_cum, (void*)l->d_tmp}) if (p) (void)hipFree(p);
    delete l;
}

int bvg_labels_info(const bvg_labels* l, int* kind, int* width, int64_t* nodes, uint64_t* stream_bytes) {
    if (!l) return BVG_E_ARG;
    if (kind) *kind = l->kind;
    if (width) *width = l->width;
    if (nodes) *nodes = l->nodes;
    if (stream_bytes) *stream_bytes = l->nbytes;
    return 0;
}

// Labels of the arcs of nodes [from,to), in the order bvg_decode_range lists the successors: d_outdeg[to-from] (int32, device)
// -> d_labels (int32, device).  *n_labels = sum of the outdegrees; BVG_E_CAPACITY if cap is smaller (nothing is written).
int bvg_labels_decode_range_dev(bvg_labels* l, int64_t from, int64_t to, const void* d_outdeg, void* d_labels, uint64_t cap, uint64_t* n_labels) {
    if (!l || from < 0 || to < from || to > l->nodes || (to > from && !d_outdeg)) return BVG_E_ARG;
    if (l->kind == BVG_LABEL_FIXED_INT_LIST || l->kind == BVG_LABEL_FIXED_LONG_LIST) return BVG_E_UNSUPPORTED;   // use bvg_labels_decode_range_lists[64]
    LCHK(hipSetDevice(l->device));
    const int64_t cnt = to - from;
    if (n_labels) *n_labels = 0;
    if (cnt == 0) return 0;
    if ((size_t)(cnt + 1) > l->deg_cap) {
        if (l->d_cum) { (void)hipFree(l->d_cum); l->d_cum = nullptr; }
        l->deg_cap = 0;
        LCHK(hipMalloc(&l->d_cum, (size_t)(cnt + 1) * sizeof(uint64_t)));
        l->deg_cap = (size_t)(cnt + 1);
    }
    const size_t tneed = scan_tmp_elems(cnt);
    if (tneed > l->tmp_cap) {
        if (l->d_tmp) { (void)hipFree(l->d_tmp); l->d_tmp = nullptr; }
        l->tmp_cap = 0;
        LCHK(hipMalloc(&l->d_tmp, tneed * sizeof(uint64_t)));
        l->tmp_cap = tneed;
    }
    launch_exclusive_scan(static_cast<const int32_t*>(d_outdeg), l->d_cum, cnt, l->d_tmp, l->stream);
    uint64_t total = 0;
    LCHK(hipMemcpyAsync(&total, l->d_cum + cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, l->stream));
    LCHK(hipStreamSynchronize(l->stream));
    if (n_labels) *n_labels = total;
    if (total > cap || (total && !d_labels)) return BVG_E_CAPACITY;
    LCHK(hipMemsetAsync(l->d_err, 0, sizeof(unsigned), l->stream));
    const uint64_t limit = l->padded - 16;
    hipLaunchKernelGGL(labels_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, l->stream, l->d_stream, limit, l->d_offsets, from, cnt,
                       static_cast<const int32_t*>(d_outdeg), l->d_cum, l->kind, l->width, static_cast<int32_t*>(d_labels), l->d_err);
    unsigned herr = 0;
    LCHK(hipMemcpyAsync(&herr, l->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, l->stream));
    LCHK(hipStreamSynchronize(l->stream));
    if (herr) return BVG_E_EOF;                                               // the degrees do not match the label stream
    return 0;
}

// Same with host buffers: outdeg[to-from] as returned by bvg_decode_range, labels[cap].
int bvg_labels_decode_range(bvg_labels* l, int64_t from, int64_t to, const int32_t* outdeg, int32_t* labels, uint64_t cap, uint64_t* n_labels) {
    if (!l || from < 0 || to < from || to > l->nodes || (to > from && !outdeg)) return BVG_E_ARG;
    LCHK(hipSetDevice(l->device));
    const int64_t cnt = to - from;
    if (n_labels) *n_labels = 0;
    if (cnt == 0) return 0;
    uint64_t total = 0;
    for (int64_t i = 0; i < cnt; i++) { if (outdeg[i] < 0) return BVG_E_ARG; total += (uint64_t)outdeg[i]; }
    if (n_labels) *n_labels = total;
    if (total > cap || (total && !labels)) return BVG_E_CAPACITY;
    int32_t* d_deg = nullptr; int32_t* d_lab = nullptr;
    LCHK(hipMalloc(&d_deg, (size_t)cnt * sizeof(int32_t)));
    if (hipMalloc(&d_lab, (size_t)(total ? total : 1) * sizeof(int32_t)) != hipSuccess) { (void)hipFree(d_deg); return BVG_E_NOMEM; }
    int r = BVG_E_HIP;
    if (hipMemcpy(d_deg, outdeg, (size_t)cnt * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess) {
        uint64_t n2 = 0;
        r = bvg_labels_decode_range_dev(l, from, to, d_deg, d_lab, total, &n2);
        if (r == 0 && total && hipMemcpy(labels, d_lab, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) r = BVG_E_HIP;
    }
    (void)hipFree(d_lab); (void)hipFree(d_deg);
    return r;
}

// List labels (FixedWidthIntListLabel): list_off[arcs+1] = exclusive prefix of the list lengths of the arcs of [from,to) in
// successor order, values = the concatenated elements.  *n_values = total element count; BVG_E_CAPACITY if cap is smaller (list_off
// is filled either way, so the caller can size the buffer and call again).  Host buffers.
int bvg_labels_decode_range_lists(bvg_labels* l, int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, int32_t* values, uint64_t cap, uint64_t* n_values) {
    return labels_lists_impl<int32_t>(l, BVG_LABEL_FIXED_INT_LIST, from, to, outdeg, list_off, values, cap, n_values);
}
// the same for FixedWidthLongListLabel: 64-bit elements
int bvg_labels_decode_range_lists64(bvg_labels* l, int64_t from, int64_t to, const int32_t* outdeg, uint64_t* list_off, int64_t* values, uint64_t cap, uint64_t* n_values) {
    return labels_lists_impl<int64_t>(l, BVG_LABEL_FIXED_LONG_LIST, from, to, outdeg, list_off, values, cap, n_values);
}

}  // extern "C"
