// bvg_api.hip — host side of libbvgraph_hip.so: the C ABI of include/bvgraph_hip.h (round 6: the entry points only; plan, index and tier scheduler live in
// bvg_plan.hip, bvg_index_host.hip and bvg_sched.hip, what they share in bvg_host.h).
//
// Mirrors the load path of the reference (ImmutableGraph.load -> BVGraph.loadInternal,
// BVGraph.java:1479-1574): parse .properties, bring .graph into memory (here: HBM), decode the
// .offsets gaps into an index (here: a device array instead of an Elias-Fano list), then serve
// outdegree / successors / sequential scans — all of which run as HIP kernels (bvg_kernels.hip).
// There is no CPU decode path in this library.
#include "bvg_host.h"

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
    for (auto& pd : g->pred2) if (pd.d_lists) (void)hipFree(pd.d_lists);
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

