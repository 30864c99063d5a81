// bvg_sched.hip — run_decode: the tier scheduler of a scan / decode of a node range (split off csrc/bvg_api.hip in round 6; see bvg_host.h).
//
// The blocks of the plan are predicted into {tier 0, four LDS size classes, giants} by the largest list they hold and launched side by side; what a tier refuses goes to
// the next (learned per block, so later scans launch it there); validated blocks run the lean scan kernel (bvg_scan.hip), the others the checking row kernel.
#include "bvg_host.h"

namespace bvghost {

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
    if (!batch && rows_default && g->skip_mode == 0 && !knob("BVG_NOSKIP") && g->tun.no_index != 1 && (to - from) >= 4096 && nblocks) {
        std::shared_ptr<SkipIndex> cur = std::atomic_load(&plp->skip);
        bool covered = cur && cur->covers(lo, lo + nblocks), retry = false;
        // a build that failed for want of memory is tried again every kRetryEvery-th scan of its blocks; so is the whole-graph rebuild behind a good partial index
        // (the countdown is shared by every handle of the graph: a compare-exchange, so that two threads at 1 cannot wrap it)
        auto tick = [](const SkipIndex& ix) { uint32_t b = ix.backoff.load(); while (b > 0 && !ix.backoff.compare_exchange_weak(b, b - 1)) {} return b; };   // the value before the tick
        // (round 6: the cause is the covering RANGE's own, and only a call that could rebuild -- a scan, or a materialising call of a quarter of the graph -- counts down)
        const bool can_build = !materialise || (to - from) >= sh->p.nodes / 4;
        if (cur && covered && cur->failed && cur->cause_of(lo, lo + nblocks) == SkipIndex::kResources) { if (can_build && tick(*cur) <= 1) { covered = false; retry = true; } }
        else if (cur && !covered && !cur->failed && can_build && tick(*cur) > 0) covered = true;
        if (!covered && (!materialise || (to - from) >= sh->p.nodes / 4)) {
            bool scanned = false;
            r = materialise ? build_skip(g, plp, 0, pl.nblk, retry) : build_skip(g, plp, lo, lo + nblocks, retry, res, from, to, &scanned);
            if (r) return r;
            if (scanned) return 0;                              // the validating pass of the build WAS this scan (same nodes, the checking kernels: bit-exact by construction)
        }
    }
    std::shared_ptr<SkipIndex> skx0 = g->skip_mode >= 2 ? g->skip_building : (g->skip_mode == 1 ? std::shared_ptr<SkipIndex>() : std::atomic_load(&plp->skip));
    if (skx0 && g->skip_mode == 0 && (skx0->failed || g->tun.no_index == 1)) skx0.reset();                // a failed build left no arrays; bvg_tuning.no_index: this handle scans without it
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
        if (flow && !is_class && aa.work_list == g->pred2[materialise ? 1 : 0].d_lists) { launch_flow_scan(aa, nb, g->flow_waves, g->flow_ws, flow_ring, st); return; }
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
        auto launch_lean = [&](DecodeArgs& al, uint32_t nb, int occ, hipStream_t st) {                 // occ: wavefronts per SIMD the instantiation leaves registers for (4: 128 VGPRs, 5: 96, 6: 80)
            if (flat_on) { al.flat_recs = flat_recs; launch_flat_decode(al, nb, occ == 6, st); }
            else launch_scan_decode(al, nb, wide, occ, materialise, st);
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
                for (uint64_t w : {24ull, 20ull, 18ull, 16ull, 14ull, 12ull, 10ull, 8ull, 6ull, 4ull}) {
                    // (round 6: 18 wavefronts -- the 96-VGPR instantiation, 4.5 per SIMD -- for graphs of 16 ... 48 arcs per node: uk +5.8 %, profiles/r06_ab_w18_*.txt; eu15, 86 arcs per node: -3 %)
                    if (w > 16 && wforce != w && !(w == 24 && avg <= 16.0 && sh->p.window_size > 0 && !wforce && !materialise)
                        && !(w == 18 && avg > 16.0 && avg <= 48.0 && sh->p.window_size > 0 && !wforce && !materialise && !knob("BVG_NO_W18"))) continue;   // more than 16: the 85-VGPR instantiation, sparse graphs with references only (web shape: +7 %; eu15: -11 % at 20; w0, all residuals: -6 %)
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
            bvg_graph::Pred& pd = g->pred2[materialise ? 1 : 0];
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
            if (tier0_first && pd.count[7]) { DecodeArgs a7 = af; a7.work_list = pd.d_lists + offc[7]; launch_lean(a7, pd.count[7], lean_waves > 20 ? 6 : (lean_waves > 16 ? 5 : 4), g->stream); launches++; alone(g->stream); }
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
                launch_lean(ac, pd.count[7 + c], 4, side_of(c)); alone(side_of(c));
                launches++;
            }
            if (t0_waits) HIPCHK(hipStreamWaitEvent(g->stream, g->side_ev[0], 0));
            if (!tier0_first && pd.count[7]) { DecodeArgs a7 = af; a7.work_list = pd.d_lists + offc[7]; launch_lean(a7, pd.count[7], lean_waves > 20 ? 6 : (lean_waves > 16 ? 5 : 4), g->stream); launches++; alone(g->stream); }
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
                bvg_graph::Pred& pd = g->pred2[materialise ? 1 : 0];
                std::sort(again.begin(), again.end());
                for (uint32_t id : work)
                    if (id < pd.learned.size() && !std::binary_search(again.begin(), again.end(), id)) { pd.learned[id] = (uint8_t)(c + 1); pd.dirty = true; }
            }
        }
        work.swap(rest);
        if (predicted_run) { bvg_graph::Pred& pd = g->pred2[materialise ? 1 : 0]; for (uint32_t id : work) if (id < pd.learned.size()) { pd.learned[id] = 5; pd.dirty = true; } }
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
        if (predicted_run) { bvg_graph::Pred& pd = g->pred2[materialise ? 1 : 0]; for (uint32_t id : work) if (id < pd.learned.size()) { pd.learned[id] = 6; pd.dirty = true; } }
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


}  // namespace bvghost
