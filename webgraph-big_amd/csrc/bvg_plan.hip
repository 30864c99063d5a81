// bvg_plan.hip — host side: parameters and handles, the block plan, the packed offsets index, opening a graph (split off csrc/bvg_api.hip in round 6; see bvg_host.h).
//
// Mirrors the load path of the reference (ImmutableGraph.load -> BVGraph.loadInternal, BVGraph.java:1479-1574): bring .graph into memory (here: HBM), turn the
// .offsets gaps into an index (here: a packed device array instead of an Elias-Fano list), cut the node range into blocks of one wavefront each.
#include "bvg_host.h"

namespace bvghost {


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


int open_common(const bvg_params* p, const uint8_t* h_graph, const void* d_graph_in, uint64_t nbytes, const uint64_t* h_offsets,
                const void* d_offsets_in, int device, bvg_graph** out, const PackedOffsets* packed) {
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


}  // namespace bvghost
