// bvg_index_host.hip — host side of the residual skip index: its granularity, the build (counting pass, dense walk, validating pass = the first scan), the failure
// bookkeeping, and basename.bvgidx on disk (split off csrc/bvg_api.hip in round 6; see bvg_host.h).  The kernels are in bvg_index.hip.
#include "bvg_host.h"

namespace bvghost {

// The granularity of a graph's skip index: lists of >= `smin` residuals hold one entry per 2^shift residuals.  A residual pass lasts as long as its longest task, so the
// threshold matters as much as the spacing: 16 / 16 (a list of 16-23 residuals is two tasks instead of one of up to 23 steps) gains on every shape over rounds 1-3's 24 / 16
// -- w0 +7.2 %, uk +3.8 %, web +2.6 %, eu +1.5 %, eu15 +1.0 % -- for 0.1-8 % more entries.  A sparse graph's pass holds few tasks, and one entry per 8 residuals from
// lists of 8 on shortens it further: web +11.5 %, uk +5.1 %, cnr-2000 +2.3 % over 24 / 16, for 0.1-0.3 GB of entries per GB of stream; on the dense default workload 8 / 8
// is no faster than 16 / 16 and takes +80 % of an index that is half the stream already, on the reference-free w0 neither (its lists are residuals only: +30 % of resident
// bytes) -- profiles/r04_skipgran3.txt.  So: 8 / 8 below 40 arcs per node (128 bits per node when the arc count is unknown) when the graph has references, else 16 / 16.
// BVG_SKIP_GRAN="min,every" (test knob) overrides; the kernels take the granularity from the index they are handed (DecodeArgs::skip_min / skip_shift), the file carries it.
void skip_granularity(const Shared* sh, uint32_t& smin, uint32_t& shift) {
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
int build_skip(bvg_graph* g, const std::shared_ptr<Plan>& plp, uint32_t blo, uint32_t bhi, bool retry_failed, bvg_scan_result* first_scan, int64_t sfrom, int64_t sto, bool* first_scan_done) {
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
    // bvg_tuning.no_index = 2 ("marks only", round 6): the validating pass and its marks -- one byte per block, what lets the lean scan kernel take the block -- but entries
    // only for lists of 4 096 residuals and more (one per 64: the giant kernel's lists, which it cannot walk in step at any useful rate): an index of ~0.03 % of the stream
    if (g->tun.no_index == 2) { ix->skip_min = 4096; ix->skip_shift = 6; }
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
            if (cur) for (const auto& r : cur->failed_ranges) if (!(blo <= r.lo && r.hi <= bhi) && fx->failed_ranges.size() < 63) fx->failed_ranges.push_back(r);   // the ranges that failed before stay failed (at most 64 with this one)
            fx->failed_ranges.push_back(SkipIndex::FailedRange{blo, bhi, cause});
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

int save_index_impl(bvg_graph* g, const char* path) {
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
int load_index_impl(bvg_graph* g, const char* path) {
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

}  // namespace bvghost
