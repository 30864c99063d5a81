// bvg_scan.hip — the lean scan kernel: tier 0 (and the big-LDS classes) of a steady-state successor scan.
//
// One wavefront per node block, everything a record touches in LDS, like the row kernel of bvg_rows.hip -- but specialised for what a
// scan of an already VALIDATED block needs (bvg_api.hip build_skip: the index-building pass of the row kernel has decoded the block
// from end to end with every consistency check on, and the stream in HBM is immutable afterwards):
//   * 32-bit successors, BVGraph's default codings (gamma / unary / zeta_k), scan mode only, residual skip index present;
//   * SUPER-ROWS and SUB-ROWS.  Up to 64 consecutive records (what fits the stream window) are parsed at once, one per lane: outdegree,
//     reference, copy blocks (straight into prefix form), intervals -- the lock-step part of the work, done with every lane busy.  All
//     references of the super-row are then known (and those of the next one's first W records are peeked at), so exactly the lists
//     that some later node copies from are STORED; every other node is a LEAF.  The lists are built in sub-rows of as many nodes as the
//     pool holds next to the stored lists of the W nodes before them -- so the LDS footprint follows the pool, not the 64-record
//     parse, and 16 wavefronts fit a CU where the row kernel holds 8.  Lists (bottom up), the parked residuals of a sub-row (top down)
//     and the super-row's copy blocks and intervals (at the very top, exactly as many as it has) share ONE area;
//   * the three streams of a record are disjoint (BVG:1062-1090 merges them; a stream where they overlap failed validation and stays
//     on the checking kernels), so nothing has to be located to be COUNTED: every residual is folded into the checksum the moment it is
//     decoded, and a leaf (~60 % of the nodes of a web graph) is never materialised, never positioned, its residuals never parked;
//   * what a leaf still owes the checksum -- the kept elements of the list it references (MaskedLongIterator.java:73-100) and its
//     interval elements (LongIntervalSequenceIterator.java:57-78) -- are RUNS: a kept copy block is a contiguous run of the referenced
//     list in LDS (read off the prefix form of the blocks), an interval is an iota.  The runs of a sub-row are cut into chunks of kChunk
//     elements dealt to all 64 lanes; a chunk is straight-line groups of {4 LDS reads in flight, 4 mixes}: no per-element block
//     bookkeeping, no branch inside a chunk;
//   * stored lists are built as in the row kernel (emission by output position, level by level; block ends and residual positions
//     are handled without branches in the position loop) -- except a stored list without reference, which is its residuals and its
//     intervals: the residuals are decoded straight into their places (nothing parked, no level, no position task), each one shifted by
//     the intervals below it, which a decoding task learns as its values pass their left ends; whoever passes an interval records where
//     it starts, and the extras pass of level 0 fills the intervals in;
//   * which of the last W lists of a super-row the NEXT super-row copies from is read off that one's records: from the staged window
//     when they lie in it, else from 12 bytes fetched from memory while the intervals are parsed;
//   * flat tasks (residual segments, extras, position tasks, chunks) are dealt to lanes by a binary search over the prefix sums of the
//     task counts with six shuffles (task_owner) -- no task maps in LDS, no loop over a lane's tasks.
// At 14-16 wavefronts per CU the kernel is bound by vector-instruction issue (profiles/r03_eu15_pmc_summary.txt, r03_ab_dummy.txt: an
// added VALU instruction costs exactly its issue time): hiding more latency (prefetching the next window or the skip entries into
// registers, 16 wavefronts with the pool they leave) does not move it; fewer instructions per arc do.
// Builds for measuring: -DBVG_PROF (wave-cycles per section), -DBVG_PROF -DBVG_PROF_WORK (`make work`: passes / steps / elements of
// every loop instead), -DBVG_MARKS (section boundaries as comments in `hipcc -S` output), -DBVG_ABLATE_{Z1,Z2,Z2LOOP,RESLOOP,LEAF,
// LEAFLOOP} (a section compiled out: wrong results, its share of the time), -DBVG_EXP_DUMMY=n (n VALU instructions added per loop step).
//
// A block that does not fit (pool, window, scratch) fails over to the row kernel's tiers like any other; a block without the
// validation mark never gets here (the host sorts on it).
#include "bvg_rows_common.h"

#include <type_traits>

#ifndef BVG_SCAN_WAVES
#define BVG_SCAN_WAVES 4
#endif

namespace bvg {

namespace {

using namespace rows;

#ifndef BVG_SCAN_CHUNK
#define BVG_SCAN_CHUNK 8
#endif
constexpr uint32_t kNoList = 0xFFFFu;
// keeps a short wave-uniform `if` a branch: the raised phi-folding thresholds of the Makefile would turn it into selects that every step pays for
// (an empty volatile asm cannot be speculated; tests/emu strips it)
#define BVG_KEEP_BRANCH(x) asm volatile("" : "+v"(x));
#ifndef BVG_SCAN_MINTASK
#define BVG_SCAN_MINTASK 1
#endif
constexpr uint32_t kScanMinTask = BVG_SCAN_MINTASK;   // shortest position task: 1 (a level of a sparse graph holds ~70 positions: one or two per lane and no loop to speak of; 4 cost the web shape 6 %)
constexpr uint32_t kChunk = BVG_SCAN_CHUNK;    // leaf elements per lane and pass (a kept copy block of a web graph is ~9 elements long)

typedef uint32_t T;

#ifdef BVG_EXP_ZTAB
// EXPERIMENT (-DBVG_EXP_ZTAB, see bvg_scan_steps3.inc): zeta_3 codes of up to 12 bits by their first 12 bits -> (value + 1) << 8 | length, 0 = longer; built at compile time
struct Zeta3Tab { uint32_t e[4096]; };
constexpr Zeta3Tab make_zeta3_tab() {
    Zeta3Tab t{};
    for (uint32_t p = 0; p < 4096; p++) {
        uint32_t h = 0; while (h < 12 && !((p >> (11 - h)) & 1u)) h++;
        uint32_t ent = 0;
        if (h <= 2) {
            const uint32_t nb = 3 * h + 3, len_long = h + 1 + nb;                         // the 3h + 3 bits behind the unary prefix
            if (len_long <= 12) {
                const uint32_t u = (p >> (12 - len_long)) & ((1u << nb) - 1u), thr = 1u << (3 * h + 1);
                const bool sh = u < thr;
                ent = ((sh ? (u + thr) >> 1 : u) << 8) | (len_long - (sh ? 1u : 0u));
            }
        }
        t.e[p] = ent;
    }
    return t;
}
__device__ const Zeta3Tab kZeta3Tab = make_zeta3_tab();
#endif
// Dealing flat tasks to lanes: every lane holds the inclusive prefix `incl` of its own task count; task t belongs to the first lane whose
// prefix exceeds t.  Six shuffles find it (a binary search over the lanes) -- no map in LDS, no loop over a lane's tasks.  EVERY lane
// must call it (a shuffle under a lane mask reads 0 from the masked lanes).
__device__ __forceinline__ uint32_t task_owner(uint32_t incl, uint32_t t) {
    // (round 6) a 4-ary search: three rounds of three INDEPENDENT shuffles (at a quarter, a half and three quarters of the span) instead of six dependent ones -- the
    // same number of vector instructions, half the chain of LDS-crossbar round trips
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = 16; step; step >>= 2) {
        const uint32_t v1 = (uint32_t)__shfl((int)incl, (int)(lo + step - 1), 64), v2 = (uint32_t)__shfl((int)incl, (int)(lo + 2 * step - 1), 64), v3 = (uint32_t)__shfl((int)incl, (int)(lo + 3 * step - 1), 64);
        lo += (v1 <= t ? step : 0u) + (v2 <= t ? step : 0u) + (v3 <= t ? step : 0u);     // (the prefixes are non-decreasing: the three tests are a thermometer)
    }
    return lo < 64 ? lo : 63u;
}
// Where the k-th set bit of a lane mask is: every lane whose bit is set PUSHES its number to the lane of its rank (ds_permute: one trip through the crossbar); lane k then
// holds the answer for k.  Lanes without a bit push to lane 63, which is read only when all 64 bits are set -- and then no such lane exists.  EVERY lane must call it.
__device__ __forceinline__ uint32_t ranked_lanes(uint64_t m) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const bool mine = (m >> lane) & 1ull;
    return (uint32_t)__builtin_amdgcn_ds_permute((int)((mine ? rk : 63u) << 2), (int)lane);
}
// The same when only a FEW lanes own tasks (the lists of one level of a sub-row: 4-8): a scalar loop over those lanes -- a v_readlane, a compare
// and a select each -- instead of six dependent trips through the LDS crossbar.  `m` = lanes that own tasks, `first` = the tasks of the
// lanes before this one (exclusive prefix); task t belongs to the last such lane whose first task is <= t.
__device__ __forceinline__ uint32_t deal_few(uint64_t m, uint32_t incl, uint32_t first, uint32_t t) {
    if (__popcll(m) > 12) return task_owner(incl, t);
    uint32_t own = threadIdx.x;
    while (m) {
        const uint32_t j = (uint32_t)__ffsll((unsigned long long)m) - 1u; m &= m - 1ull;
        const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)first, (int)j);
        own = t >= f ? j : own;
    }
    return own;
}
// gamma from the LDS window: codes of up to 31 bits (values below 2^15: every copy block and interval of a list that fits LDS) from one
// 32-bit window; the 64-bit decoder only where a lane needs it.  Returns the length, 0 = does not fit.
__device__ __forceinline__ uint32_t gamma_at(const uint32_t* stage, uint32_t rel, uint64_t& v) {
    const uint32_t w = win32<LIN>(stage, rel);
    const uint32_t lz = w ? (uint32_t)__builtin_clz(w) : 32u;
    if (lz < 16) { const uint32_t len = 2 * lz + 1; v = (w >> (32u - len)) - 1u; return len; }
    return gamma64(win64<LIN>(stage, rel), v);
}
// zeta_3 (BVGraph's default residual code) from a 32-bit window, straight-line and without the two integer multiplies of the general
// form (quarter rate): codes of up to 31 bits (h <= 6: values below 2^21 - 1); returns the length, 0 = take the 64-bit decoder
__device__ __forceinline__ uint32_t zeta3_fast32(uint32_t w, uint32_t& val) {
    const uint32_t h = w ? (uint32_t)__builtin_clz(w) : 32u;
    const uint32_t fits = h <= 6u ? 1u : 0u, hh = fits ? h : 0u;                 // h = 7 is 32 bits with the extra bit
    const uint32_t h3 = hh + (hh << 1), nb = h3 + 2u, zt = (hh << 2) + 3u;       // payload bits, code length without the extra bit
    const uint32_t t = (w << (hh + 1u)) >> (32u - nb);
    const uint32_t left = 1u << h3;
    const bool shortc = t < left;
    const uint32_t ext = ((t << 1) | ((w >> (31u - zt)) & 1u)) - 1u;
    val = shortc ? t + left - 1u : ext;
    return fits ? zt + (shortc ? 0u : 1u) : 0u;
}
// the k-th set bit of a 64-bit mask (k < popcount)
__device__ __forceinline__ uint32_t select_bit(uint64_t m, uint32_t k) {
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t step = 32; step; step >>= 1) {
        const uint32_t c = (uint32_t)__popcll(m & ((1ull << (pos + step)) - 1ull));     // set bits below pos + step (pos + step <= 63 here)
        pos += c <= k ? step : 0u;
    }
    return pos;
}

// Z3: zeta_3 residuals (the specialised decoder)
// OCC: wavefronts per SIMD the register allocation leaves room for -- 4 (128 VGPRs: nothing spills, 16 wavefronts per CU) or 6 (85 VGPRs, a
// handful of spills, 24 per CU: sparse graphs, whose lists need little LDS, gain 9 % from the extra wavefronts; profiles/r03_ab_w20.txt)
// MAT: the materialising form behind nodeIterator() / successorBigArray() (BVG:1164-1176, a.succ / a.cum / a.outdeg): every list of a reported
// node is written to the output -- lists that are copied from and lists without reference are built in LDS exactly as in scan mode and copied
// out in coalesced runs; a LEAF with a reference (most nodes) is merged straight into the output by the same position tasks (its parked
// residuals and its block / interval entries are all it needs of LDS); nothing is summed (no checksum: the caller gets the arcs).
template <bool Z3, bool WIDE, int OCC, bool D2, bool MAT>
__global__ void __launch_bounds__(64, OCC) scan_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];     // pool | scratch | stream window

    const unsigned lane = threadIdx.x;
    const uint32_t wi = xcd_order(blockIdx.x, gridDim.x, a.xcds);                      // neighbouring blocks on one XCD (bvg_rows_common.h)
    const uint32_t bid = a.work_list ? a.work_list[wi] : (a.blk_lo + wi);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const uint32_t W = (uint32_t)a.window;
    const int64_t hs = s - (int64_t)halo;
    const int64_t rep_lo = s > a.from ? s : a.from, rep_hi = e < a.to ? e : a.to;

    T* const pool = reinterpret_cast<T*>(dyn_lds);
    // ONE area for the lists (bottom up), the parked residuals of a sub-row (top down, below the scratch) and the super-row's scratch -- copy
    // blocks and intervals, at the very top, as large as this super-row needs (a fixed scratch area stood 60 % empty on average): what the
    // scratch does not take, the sub-rows get.  `scr` indexes the same area; sb / ib are absolute.
    T* const scr = pool;
    // WW (round 6, scan mode of the dense instantiations): a stored list WITH reference is built wave-wide, one list after the other, from lane bit vectors (below, "WW");
    // the bit vector (the copy mask over the referenced list, then the positions of the extras in the list being built) takes kWWWords dwords off the top of the area
#ifdef BVG_EXPERIMENTAL
    constexpr bool WWT = !MAT && OCC <= 5;                                   // (`make experimental` and the emulator only: it lost, see below)
#else
    constexpr bool WWT = false;
#endif
    constexpr uint32_t kWWBits = 1024, kWWWords = kWWBits / 32;              // lists and referenced lists of up to 1 024 elements (longer ones keep the position tasks)
    // MEASURED (profiles/r06_ab_ww3_*.txt, r06_ww_on_eu15_2g_pmc_summary.txt): eu15 302 -> 270 G edges/s with every such list built this way, 287 / 297 / 298 G from 128 / 256 / 512
    // elements on; uk 218 -> 179.  The vector instructions do drop (1.59 -> 1.40 per arc) but the scalar ones rise (0.75 -> 1.01) and every list is a serial chain of LDS round
    // trips and scalar prefix steps (~3 000 cycles for a list of 100): vector-issue utilisation falls from 92 % to 71 % -- the wavefront's own critical path becomes the bound.
    // Opt-in in the experimental build only: BVG_DBG = 4096 + (shortest list << 16).
    const bool wwon = WWT && (a.dbg & 4096u) != 0;
    const uint32_t CAPfull = a.lds_pool_elems + a.lds_scr_elems;
    // ZE (round 6, scan mode): the position tasks of a level walk the KEPT elements only; where the extras lie in the lists being built is a bit vector per list (set by the
    // extras pass), and a copied element's place is the next clear bit.  The bit vectors of one level's lists live in kZEWords dwords off the top of the area.
    // MEASURED (profiles/r06_ab_ze_*.txt, r06_ze_on_eu15_2g_pmc_summary.txt against r06_mid_eu15_2g_pmc_summary.txt): eu15 304 -> 294 G edges/s, cnr 142 -> 137, uk 220 -> 218.
    // The position loop does shrink (no step on a residual's place, 33 instead of 43 instructions per step), but the intervals then have to be written by the extras pass, a lane
    // each and element by element -- ~25 steps of a 64-wide pass for one lane's work -- and the bit vectors cost a zeroing, an atomic per extra, a select per task and 3 % of the
    // pool: 1.556 vector instructions per arc against 1.517.  Opt-in in the experimental build only (BVG_DBG = 8192).
#ifdef BVG_EXPERIMENTAL
    const bool zeon = !MAT && (a.dbg & 8192u) != 0;
#else
    constexpr bool zeon = false;
#endif
    const uint32_t kZEWords = 32u + (CAPfull >> 6);
    const uint32_t CAP = CAPfull - (wwon ? kWWWords : 0u) - (zeon ? kZEWords : 0u);
    uint32_t* const wwm = reinterpret_cast<uint32_t*>(pool + CAP);
    uint32_t* const zem = reinterpret_cast<uint32_t*>(pool + CAP + (wwon ? kWWWords : 0u));
    uint32_t* const stage_w = reinterpret_cast<uint32_t*>(pool + CAPfull);   // the window over the stream: read by every sub-row of a super-row
    // the node ring BEHIND the window, in the dynamic allocation (round 6): with no static LDS in front of it the dynamic area starts at LDS address 0, and the byte
    // addresses the bit cursors of the hot loops compute need no base added
    uint16_t* const nd_base = reinterpret_cast<uint16_t*>(stage_w + a.lds_stage_words);    // first pool element of a node's list (kNoList: a leaf, no list); pools hold < 65535 elements
    uint16_t* const nd_d = nd_base + kRing;                                                // outdegree, clamped (a list longer than the pool fails the block before anything copies from it)
    const uint32_t* const stage = stage_w;
    const uint32_t stage_bits = a.lds_stage_words * 32u;
    const uint32_t sbitw = (uint32_t)(reinterpret_cast<const unsigned char*>(stage_w) - dyn_lds) << 3;   // the window's first bit, counted from the start of the dynamic LDS
    const uint32_t zk = (uint32_t)a.cod.zeta_k, minint = (uint32_t)a.min_interval;
    const bool zfast = zk >= 2;
    const uint32_t kSkipMin = a.skip_min, kSkipShift = a.skip_shift, kSkipEvery = 1u << kSkipShift;   // (this index's granularity: they hide the compile-time defaults of bvg_kernels.h)
    // WIDE (graphs of more than 2^32 - 256 nodes): the lists still hold 32-bit elements -- ids RELATIVE to a per-block base B, 2^31 below the
    // block's first node.  Successors of a web graph lie near their node; a block with an id outside [B, B + 2^32) fails (every id that enters
    // a list is checked where it is made: residuals, skip values, interval ends; copied elements come from checked lists) and stays with
    // the 64-bit row kernel.  The checksum takes the base like a node base (mix_node: m + base, carries included).
    const int64_t B = WIDE ? (s > (int64_t)a.wide_half ? s - (int64_t)a.wide_half : 0) : 0;
    const uint64_t nbase = a.node_base + (uint64_t)B;

    for (unsigned i = lane; i < (unsigned)kRing; i += 64) { nd_base[i] = 0; nd_d[i] = 0; }
    wave_sync();

    uint32_t pool_used = 0;
    uint64_t stg_bit0 = 0; uint32_t stg_bits = 0;             // staged window (wave-uniform)
    uint64_t blk_arcs = 0, blk_chk = 0, blk_nodes = 0;
    unsigned err = 0;
    bool failed = false;
    uint32_t fail_need = 0xFFFFFFFFu;                        // pool elements that would have been enough (when known)
    uint32_t cnt_super = 0, cnt_sub = 0, cnt_nodes = 0;      // BVG_DBG & 64: work counters (wave-uniform)
    // -DBVG_PROF builds only (`make prof`): wave-cycles per section, reported with BVG_DBG & 64 through the row kernel's counters
    // {0 descriptors + levels, 1 Z2 sizing, 2 Z2 set-up, 3 Z1, 4 Z2 loop, 5 phase 1, 6 row set-up, 7 headers, 8 pool sizing, 9 residuals, 10 leaf pass, 11 leaf loop}
#if defined(BVG_PROF) && defined(BVG_PROF_WORK)
    // `make work`: the same slots count WORK instead of cycles (wave-uniform counts; lane sums go through BVG_WCL):
    // {0 levels with members, 1 Z2 passes, 2 Z2 tasks, 3 Z1 passes, 4 Z2 loop steps, 5 Z2 positions, 6 residual task passes, 7 residual task-loop steps,
    //  8 residuals decoded by tasks, 9 steps of the lane-per-node residual loop, 10 leaf chunk passes, 11 leaf loop steps (4 elements each),
    //  12 leaf elements, 13 Z1 tasks, 14 residuals of the lane-per-node loop, 15 leaf item passes}
    uint32_t cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t cyc2[4] = {0, 0, 0, 0};                         // round 6: {0 steps of the copy-block loop (pairs), 1 steps of the interval loop, 2 copy blocks, 3 intervals}
#define BVG_T0() 0u
#define BVG_T1(i, t) do { (void)(t); } while (0)
#define BVG_WC(i, n) do { cyc[i] += (uint32_t)(n); } while (0)
#define BVG_WCL(i, n) do { cyc[i] += wave_sum32((uint32_t)(n)); } while (0)
#define BVG_WC2(i, n) do { cyc2[i] += (uint32_t)(n); } while (0)
#elif defined(BVG_MARKS)
    // `hipcc -S -DBVG_MARKS`: the section boundaries as comments in the assembly (static instruction counts per section)
#define BVG_T0() ([]() { asm volatile("; BVGMARK begin"); return 0u; }())
#define BVG_T1(i, t) do { (void)(t); asm volatile("; BVGMARK end %0" :: "n"(i)); } while (0)
#define BVG_WC(i, n) do { } while (0)
#define BVG_WCL(i, n) do { } while (0)
#define BVG_WC2(i, n) do { } while (0)
#elif defined(BVG_PROF)
    uint32_t cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define BVG_T0() ((uint32_t)clock64())
#define BVG_T1(i, t) do { cyc[i] += (uint32_t)clock64() - (t); } while (0)
#define BVG_WC(i, n) do { } while (0)
#define BVG_WCL(i, n) do { } while (0)
#define BVG_WC2(i, n) do { } while (0)
#else
#define BVG_T0() 0u
#define BVG_T1(i, t) do { (void)(t); } while (0)
#define BVG_WC(i, n) do { } while (0)
#define BVG_WCL(i, n) do { } while (0)
#define BVG_WC2(i, n) do { } while (0)
#endif

    // residual skip index: entries of this block
    const uint64_t sk_base = a.skip_first[bid];
    const uint32_t sk_n = (uint32_t)(a.skip_first[bid + 1] - sk_base);
    uint32_t sk_run = 0;

    // keeps only the stored lists of the W nodes before node `upto`, moved to the front of the pool
    auto compact = [&](int64_t upto) {
        uint32_t my_d = 0, my_base = 0; const int64_t y = upto - (int64_t)W + (int64_t)lane;
        const bool livelane = lane < W && y >= hs;
        if (livelane) { my_base = nd_base[(uint32_t)y & RM]; my_d = my_base == kNoList ? 0u : nd_d[(uint32_t)y & RM]; }
        const uint32_t nincl = wave_incl_scan32(my_d);
        const uint32_t nbase = nincl - my_d;
        for (uint32_t jn = 0; jn < W && jn < 64; jn++) {
            const uint32_t src = lane_get(my_base, jn), dst = lane_get(nbase, jn), len = lane_get(my_d, jn);
            if (src != dst && len)
                for (uint32_t t0 = 0; t0 < len; t0 += 64) {   // (the lists move down over themselves 64 elements at a time: every element of a step is read before any is written)
                    const uint32_t t = t0 + lane; T vv = 0;
                    if (t < len) vv = pool[src + t];
                    wave_sync();
                    if (t < len) pool[dst + t] = vv;
                    wave_sync();
                }
        }
        if (livelane && my_base != kNoList) nd_base[(uint32_t)y & RM] = (uint16_t)nbase;
        pool_used = lane_get(nincl, 63);
        wave_sync();
    };

    int64_t r0 = hs;
    // offsets of the first super-row (later ones are prefetched while the previous one is decoded)
    uint64_t off_x = 0, rec_end = 0;
    if (r0 + lane < e) { off_x = a.offsets[r0 + lane]; rec_end = a.offsets[r0 + lane + 1]; }

    while (r0 < e && !failed) {
        // ================================================================== SUPER-ROW: up to 64 nodes, one per lane.  Their records are
        // parsed once, with every lane busy; the lists are then built in SUB-ROWS of as many nodes as the pool holds.
        const uint32_t tq5 = BVG_T0();
        const int64_t x = r0 + lane;
        const bool in_range = x < e;
        const uint32_t hbit = x < s ? (uint32_t)(s - 1 - x) : 0;
        const bool needed = in_range && (x >= s || ((hmask >> hbit) & 1ull));
        const uint32_t left = (uint32_t)(e - r0 > 64 ? 64 : e - r0);
        { const uint32_t tqc = BVG_T0(); if (pool_used > 0) compact(r0); BVG_T1(12, tqc); }
        const uint32_t tqs = BVG_T0();
        {   // (re)stage the window when this super-row's records are not covered by it
            const uint64_t row_lo = lane_get64(off_x, 0);
            const uint64_t row_hi = lane_get64(rec_end, left - 1);
            if (!(row_lo >= stg_bit0 && row_hi + 96 <= stg_bit0 + stg_bits)) {
                wave_sync();
                const uint64_t b0 = (row_lo >> 3) & ~15ull;
                uint64_t nb = a.padded_bytes > b0 ? a.padded_bytes - b0 : 0;
                if (nb > (stage_bits >> 3)) nb = stage_bits >> 3;
                for (uint32_t c = lane; c < (uint32_t)(nb >> 4); c += 64) {
                    const uint4 v = *reinterpret_cast<const uint4*>(a.graph + b0 + ((uint64_t)c << 4));
                    uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
                    *reinterpret_cast<uint4*>(&stage_w[c << 2]) = w;
                }
                stg_bit0 = b0 << 3; stg_bits = (uint32_t)(nb << 3);
                wave_sync();
            }
        }
        BVG_T1(13, tqs);
        // the super-row is cut where the records stop fitting the window (the next one restages from there)
        const bool inwin = in_range && rec_end + 96 <= stg_bit0 + stg_bits && off_x >= stg_bit0;
        uint32_t K1;
        {   // contiguous prefix only
            const uint64_t m = ballot(inwin);
            K1 = m == ~0ull ? 64u : (uint32_t)__ffsll((unsigned long long)~m) - 1u;
            if (K1 > left) K1 = left;
        }
        if (K1 == 0) { failed = true; fail_need = 0xFFFFFFF1u; break; }       // a single record larger than the window
        // prefetch the next super-row's offsets now (their latency hides behind the header parse); the scratch area may still cut
        // this super-row shorter, then they are fetched again below
        const uint32_t K1win = K1;
        uint64_t nxt_off = 0, nxt_end = 0;
        {
            const int64_t nx = r0 + K1 + lane;
            if (nx < e) { nxt_off = a.offsets[nx]; nxt_end = a.offsets[nx + 1]; }
        }
        uint32_t rel = (uint32_t)(off_x - stg_bit0);                          // bit cursor relative to the window
        const uint32_t pend = (uint32_t)(rec_end - stg_bit0);
        const uint32_t recrel = rel;
        bool bad = false;
        uint64_t v;
        uint32_t d = 0;
        // (round 6) the header fields from ONE 32-bit window each -- two dword reads and a funnel shift (the running bit address of bvg_scan_steps3.inc; at bit 0 of the
        // window the shift is 0 and the dword in front of it, which is read along, drops out) -- and the 64-bit decoders only for a code of more than 31 bits
        auto win_at = [&](uint32_t r) -> uint32_t {
            const uint32_t tb0 = sbitw + r - 1u;
            const uint32_t* const wp = reinterpret_cast<const uint32_t*>(dyn_lds + ((tb0 >> 3) & ~3u));
            return __builtin_amdgcn_alignbit(wp[0], wp[1], ~tb0);
        };
        auto gamma_w32 = [&](uint32_t r, uint64_t& val) -> uint32_t {          // gamma: the length, 0 = does not fit 64 bits
            const uint32_t w = win_at(r);
            if (__builtin_expect(w < 0x10000u, 0)) return gamma64(win64<LIN>(stage, r), val);
            const uint32_t len = 2u * (uint32_t)__builtin_clz(w) + 1u;
            val = (w >> (32u - len)) - 1u;
            return len;
        };
        if (needed && lane < K1) {                                            // readOutdegree, BVG:654-660
            const uint32_t l = gamma_w32(rel, v);
            bad |= l == 0 || v > 0x7FFFFFFFull; rel += l; d = bad ? 0u : (uint32_t)v;
        }
        const uint32_t dclamp = d > CAP ? CAP + 1 : d;
        if (needed && lane < K1) nd_d[(uint32_t)x & RM] = (uint16_t)(d < 0xFFFFu ? d : 0xFFFFu);
        wave_sync();
        BVG_T1(6, tq5);
        const uint32_t tq7 = BVG_T0();
        // ------------------------------------------------------------------ phase 1: every lane parses the header of its record
#ifdef BVG_EXP_PRIO
        __builtin_amdgcn_s_setprio(3);                                        // experiment: the lock-step header parse ahead of the other wavefronts' bulk loops
#endif
        uint32_t ref = 0, bc = 0, ic = 0, nres = 0, sb = 0, ib = 0;
        int32_t extra = (int32_t)d;                                           // (32-bit arithmetic from here on: the lists of this kernel hold 32-bit ids,
        uint32_t big = 0;                                                     //  and a code value that would not fit fails the block)
        bool parse = needed && lane < K1 && d > 0;
        // ---- A: reference and block count (BVG:1015-1021)
        if (parse) {
            if (W > 0) {                                                      // readReference, BVG:692-703
                const uint32_t w3 = win_at(rel);
                uint32_t lz = w3 ? (uint32_t)__builtin_clz(w3) : 32u;
                if (__builtin_expect(w3 == 0u, 0)) { const uint64_t w = win64<LIN>(stage, rel); lz = w ? (uint32_t)__builtin_clzll(w) : 64u; }
                v = lz;
                const uint32_t l = lz < 64 ? lz + 1 : 0;
                bad |= l == 0; rel += l;
                if (v > W || (int64_t)v > x) { err |= ERR_REF_RANGE; v = 0; }
                ref = (uint32_t)v;
            }
            if (ref > 0) {                                                    // readBlockCount, BVG:728-735
                const uint32_t l = gamma_w32(rel, v);
                bad |= l == 0 || v > pend - rel + 1; rel += l; bc = bad ? 0u : (uint32_t)v;
            }
        }
        // the copy blocks and intervals of the super-row go to the scratch area
        const uint32_t SCRH = (CAP - pool_used) >> 1;                         // at most half of what the lists carried over leave free
        const uint32_t bincl = wave_incl_scan32(bc > SCRH ? SCRH + 1 : bc);
        { const uint32_t kb = (uint32_t)__popcll(ballot(bincl <= SCRH)); K1 = kb < K1 ? kb : K1; }
        if (K1 == 0) { failed = true; fail_need = 0xFFFFFFF3u; break; }       // one node's copy blocks exceed the scratch area
        const uint32_t btot = lane_get(bincl, K1 - 1);                        // (stays, should the intervals cut the super-row shorter: the blocks are written by then)
        const uint32_t bbase = CAP - btot;
        sb = bbase + bincl - bc;
        // ---- B: copy blocks (BVG:1023-1032) and C: interval count (BVG:1040)
        uint32_t rlenN = 0, ncopN = 0;                                        // the referenced list's length; elements copied from it (BVG:1030)
#if defined(BVG_PROF) && defined(BVG_PROF_WORK)
        uint32_t ncop_sim = 0;
#endif
#if defined(BVG_PROF) && defined(BVG_PROF_WORK)
        { const bool pb = parse && lane < K1 && ref > 0; BVG_WC2(0, wave_max32(pb ? bc >> 1 : 0u)); BVG_WC2(2, wave_sum32(pb ? bc : 0u)); }
#endif
        if (parse && lane < K1) {
            if (ref > 0) {
                uint32_t copied = 0, tot = 0;
                // two blocks per step -- a kept one and the skipped one behind it -- from one 64-bit window: the longest block list of the 64
                // records sets the number of steps (28 on the eu15 shape, 8 % of the scan at one block per step: profiles/r03_ab_dummyhdr.txt).
                // A block of 2^16 - 1 elements or more fails the block here (no list of this kernel is that long).
                // Round 6: both codes from ONE 32-bit window when they fit it (each of up to 31 bits, 32 together: blocks of fewer than ~2^7 elements, nearly all of
                // them) -- two dword reads and a funnel shift instead of three reads and 64-bit shifts; a lane whose pair does not fit takes the 64-bit form of the
                // step.  The loop takes PAIRS (a kept block and the skipped one behind it); the last block of an odd count follows it.  `tot` and `copied` start
                // at -1: every block but the first is stored minus 1 (BVG:1025), so adding (gamma value + 1) = the code's top bits is right for all of them.
                uint32_t tbh = sbitw + rel - 1u;                              // running bit address of the next code, minus one (as in bvg_scan_steps3.inc)
                const uint32_t tbend = sbitw + pend - 1u;
                tot = 0xFFFFFFFFu; copied = 0xFFFFFFFFu;
                uint32_t i = 0;
                for (; i + 1u < bc && tbh <= tbend; i += 2) {
                    const uint32_t* const wp = reinterpret_cast<const uint32_t*>(dyn_lds + ((tbh >> 3) & ~3u));
                    const uint32_t w = __builtin_amdgcn_alignbit(wp[0], wp[1], ~tbh);
                    const uint32_t z1 = (uint32_t)__builtin_clz(w | 0x10000u), l1 = 2u * z1 + 1u;      // <= 15 zeros, <= 31 bits
                    const uint32_t w2 = w << l1;
                    const uint32_t z2 = (uint32_t)__builtin_clz(w2 | 0x10000u), l2 = 2u * z2 + 1u;
                    uint32_t b1 = w >> (32u - l1), b2 = w2 >> (32u - l2);     // gamma value + 1 = the top bits of the code
                    uint32_t adv = l1 + l2;
                    if (__builtin_expect((w < w2 ? w : w2) < 0x10000u || adv > 32u, 0)) {
                        const uint64_t w6 = win64<LIN>(stage, tbh + 1u - sbitw);
                        const uint32_t lz1 = w6 ? (uint32_t)__builtin_clzll(w6) : 64u;
                        const uint32_t m1 = 2u * (lz1 & 15u) + 1u;
                        const uint64_t w62 = w6 << m1;
                        const uint32_t lz2 = w62 ? (uint32_t)__builtin_clzll(w62) : 64u;
                        const uint32_t m2 = 2u * (lz2 & 15u) + 1u;
                        bad |= lz1 >= 16u || lz2 >= 16u;                      // a block of 2^16 - 1 elements or more fails the block here (no list of this kernel is that long)
                        b1 = (uint32_t)(w6 >> (64u - m1)); b2 = (uint32_t)(w62 >> (64u - m2));
                        adv = m1 + m2;
                    }
                    tot += b1; copied += b1;
                    // straight in PREFIX form (MaskPrefix): end position of block i in the referenced list | elements kept up to and including it
                    const T e1 = MaskPrefix<T>::pack(tot, copied);
                    tot += b2;
                    scr[sb + i] = e1; scr[sb + i + 1u] = MaskPrefix<T>::pack(tot, copied);
                    tbh += adv;
#ifdef BVG_EXP_DUMMY_HDR
                    { uint32_t dm = i; _Pragma("unroll") for (int z = 0; z < BVG_EXP_DUMMY_HDR; z++) asm volatile("v_xad_u32 %0, %0, %0, %0" : "+v"(dm)); if (dm == 0x12345u) err |= 1u; }
#endif
                }
                if (i < bc && tbh <= tbend) {                                 // the last block of an odd number
                    const uint64_t w6 = win64<LIN>(stage, tbh + 1u - sbitw);
                    const uint32_t lz1 = w6 ? (uint32_t)__builtin_clzll(w6) : 64u;
                    const uint32_t m1 = 2u * (lz1 & 15u) + 1u;
                    bad |= lz1 >= 16u;
                    const uint32_t b1 = (uint32_t)(w6 >> (64u - m1));
                    tot += b1; copied += b1;
                    scr[sb + i] = MaskPrefix<T>::pack(tot, copied);
                    tbh += m1; i++;
                }
                bad |= i < bc;                                                // (the loop stops at the record's end: blocks left over are a malformed record)
                if (bc == 0) { tot = 0; copied = 0; }
                rel = tbh + 1u - sbitw;
                rlenN = nd_d[(uint32_t)(x - ref) & RM];
                if (big != 0 || tot > rlenN || tot > 0xFFFFu) { bad = true; tot = rlenN; }   // (cannot happen in a validated block)
                if (!(bc & 1)) copied += rlenN - tot;                         // BVG:1030
                extra = (int32_t)d - (int32_t)copied; ncopN = copied;
#if defined(BVG_PROF) && defined(BVG_PROF_WORK)
                ncop_sim = copied;
#endif
                if (extra < 0) bad = true;
            }
            if (extra > 0 && minint != 0) {                                   // always gamma
                const uint32_t l = gamma_w32(rel, v);
                bad |= l == 0 || v > (pend - rel) / 2 + 1; rel += l; ic = bad ? 0u : (uint32_t)v;
            }
        }
        // The next super-row's first W records decide which of this one's last lists are copied from (see below); they mostly lie
        // behind the staged window, so their first 12 bytes are fetched from memory here and looked at after the intervals.
        uint32_t pk0 = 0, pk1 = 0, pk2 = 0; bool peeked = false;
        {
            const int64_t nx = r0 + K1 + lane;
            const uint64_t pb = (nxt_off >> 5) << 2;
            if (K1 == K1win && lane < W && nx < e && !(nxt_off >= stg_bit0 && nxt_off + 160 <= stg_bit0 + stg_bits) && pb + 12 <= a.padded_bytes) {
                const uint32_t* gp = reinterpret_cast<const uint32_t*>(a.graph + pb);
                pk0 = gp[0]; pk1 = gp[1]; pk2 = gp[2]; peeked = true;
            }
        }
        const uint32_t iw = lane < K1 ? 2 * ic : 0u;
        const uint32_t iincl = wave_incl_scan32(iw > SCRH ? SCRH + 1 : iw);
        // the copy blocks may have taken the whole area: halve the super-row until the first node's intervals fit
        for (;;) {
            const uint32_t ki = (uint32_t)__popcll(ballot(btot + iincl <= SCRH));
            if (ki != 0 || K1 <= 1) { K1 = ki < K1 ? ki : K1; break; }
            K1 = (K1 + 1u) >> 1;
        }
        if (K1 == 0) { failed = true; fail_need = 0xFFFFFFF4u; break; }       // one node's intervals exceed the scratch area
        const uint32_t itot = lane_get(iincl, K1 - 1);
        const uint32_t CAPe = bbase - itot;                                   // what is left for the lists and the parked residuals of this super-row's sub-rows
        ib = CAPe + iincl - iw;
        parse = parse && lane < K1;
        // ---- D1: intervals (BVG:1042-1058): they fix the number of residuals
#if defined(BVG_PROF) && defined(BVG_PROF_WORK)
        { BVG_WC2(1, wave_max32(parse ? ic : 0u)); BVG_WC2(3, wave_sum32(parse ? ic : 0u)); }
#endif
        if (parse) {
            if (ic > 0) {
                uint32_t prev = 0;
                // Round 6: left end and length from ONE 32-bit window when the two gamma codes fit it (as in the copy-block loop above)
                uint32_t tbi = sbitw + rel - 1u, negacc = 0;
                const uint32_t tiend = sbitw + pend - 1u;
                uint32_t i = 0;
                for (; i < ic && tbi <= tiend; i++) {
                    const uint32_t* const wp = reinterpret_cast<const uint32_t*>(dyn_lds + ((tbi >> 3) & ~3u));
                    const uint32_t w = __builtin_amdgcn_alignbit(wp[0], wp[1], ~tbi);
                    const uint32_t z1 = (uint32_t)__builtin_clz(w | 0x10000u), l1 = 2u * z1 + 1u;
                    const uint32_t w2 = w << l1;
                    const uint32_t z2 = (uint32_t)__builtin_clz(w2 | 0x10000u), l2 = 2u * z2 + 1u;
                    uint32_t u1 = (w >> (32u - l1)) - 1u, u2 = (w2 >> (32u - l2)) - 1u;      // the two gamma values
                    uint64_t v1 = u1;                                         // (WIDE only: the left gap at full width)
                    uint32_t adv = l1 + l2;
                    if (__builtin_expect((w < w2 ? w : w2) < 0x10000u || adv > 32u, 0)) {
                        const uint32_t r1 = tbi + 1u - sbitw;
                        uint64_t v2;
                        const uint32_t m1 = gamma_at(stage, r1, v1);
                        const uint32_t m2 = gamma_at(stage, r1 + m1, v2);
                        if (m1 == 0 || m2 == 0) { bad = true; v1 = 0; v2 = 0; }
                        big |= (uint32_t)(v1 >> 32) | (uint32_t)(v2 >> 23) | (uint32_t)(v2 >> 32);   // (a run descriptor holds 24 bits of length: leave a longer interval to the row kernel)
                        u1 = (uint32_t)v1; u2 = (uint32_t)v2;
                        adv = m1 + m2;
                    }
                    tbi += adv;
#ifdef BVG_EXP_DUMMY_IV
                    { uint32_t dm = i; _Pragma("unroll") for (int z = 0; z < BVG_EXP_DUMMY_IV; z++) asm volatile("v_xad_u32 %0, %0, %0, %0" : "+v"(dm)); if (dm == 0x12345u) err |= 1u; }
#endif
                    uint32_t leftv = prev + 1u + u1;
                    if (i == 0) {                                             // (wave-uniform) the first left end is relative to the node, signed (BVG:1047)
                        uint32_t l0 = (uint32_t)(x - B) + ((u1 >> 1) ^ (0u - (u1 & 1u)));    // nat2int, modulo 2^32
                        BVG_KEEP_BRANCH(l0);
                        leftv = l0;
                    }
                    const uint32_t len = u2 + minint;
                    if (WIDE) {                                               // the interval must lie inside the block's 2^32 ids
                        const int64_t lt = i == 0 ? (x - B) + nat2int64(v1) : (int64_t)(uint64_t)prev + 1 + (int64_t)v1;
                        big |= (uint32_t)(((uint64_t)lt) >> 32) | (uint32_t)(((uint64_t)lt + len) >> 32);
                    }
                    prev = leftv + len;
                    extra -= (int32_t)len;
                    negacc |= (uint32_t)extra;                                // (checked at every step: the difference must not wrap)
                    scr[ib + 2 * i] = (T)leftv; scr[ib + 2 * i + 1] = (T)len;
                }
                bad |= i < ic || (negacc >> 31) != 0u;                                                // (stopped at the record's end with intervals left over)
                rel = tbi + 1u - sbitw;
                if (extra < 0 || big != 0) { bad = true; extra = 0; }
            }
            nres = (uint32_t)extra;
        }
          // the touches have landed long ago; their registers are free again
#ifdef BVG_EXP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (ballot(bad && lane < K1)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
        BVG_T1(7, tq7);
        const uint32_t tq8 = BVG_T0();
        // ---- which lists are STORED: those that a later node copies from -- referenced inside the super-row (every reference of the
        //      super-row is known now) or one of its last W nodes, which the next super-row may reference.  Every other node is a leaf:
        //      no list, no parked residuals, only run descriptors.
        const bool on1 = needed && lane < K1;
        const uint32_t tqp = BVG_T0();
        uint64_t refmask = 0;
        for (uint32_t r = 1; r <= W && r < 64; r++) refmask |= ballot(parse && ref == r) >> r;
        if (K1 != K1win) {
            const int64_t nx = r0 + K1 + lane;
            nxt_off = 0; nxt_end = 0;
            if (nx < e) { nxt_off = a.offsets[nx]; nxt_end = a.offsets[nx + 1]; }
        }
        // The last W nodes of the super-row can be referenced by the first W nodes of the next one: peek at those records' references
        // (outdegree gamma, reference unary: BVG:654-660, 692-703) when they lie inside the staged window; otherwise, assume they are.
        // Nodes of the next BLOCK do not count: that block decodes its halo itself.
        {
            uint32_t tgt = 64;                                                // lane of this super-row that my peeked node references
            bool unknown = false;
            const int64_t nx = r0 + K1 + lane;
            if (lane < W && nx < e) {
                unknown = true;
                if (nxt_off >= stg_bit0 && nxt_off + 160 <= stg_bit0 + stg_bits) {
                    uint32_t prel = (uint32_t)(nxt_off - stg_bit0);
                    uint64_t pv;
                    const uint32_t l = gamma64(win64<LIN>(stage, prel), pv);
                    if (l != 0) {
                        unknown = false;
                        if (pv != 0) {
                            const uint64_t w = win64<LIN>(stage, prel + l);
                            const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                            if (lz > lane && lz <= W && lz - lane <= K1) tgt = K1 + lane - lz;     // reaches back into this super-row
                            else if (lz >= 64) unknown = true;
                        }
                    }
                } else if (peeked && K1 == K1win) {                                           // the 12 bytes fetched above
                    const uint32_t sh = (uint32_t)nxt_off & 31u;
                    const uint64_t hi = ((uint64_t)__builtin_bswap32(pk0) << 32) | __builtin_bswap32(pk1);
                    const uint64_t win = sh ? (hi << sh) | (uint64_t)(__builtin_bswap32(pk2) >> (32u - sh)) : hi;
                    uint64_t pv;
                    const uint32_t l = gamma64(win, pv);
                    if (l != 0 && l <= 48) {
                        unknown = false;
                        if (pv != 0) {
                            const uint64_t w = win << l;
                            const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                            if (lz >= 64u - l) unknown = true;                                 // (the unary code runs past the 64 bits at hand)
                            else if (lz > lane && lz <= W && lz - lane <= K1) tgt = K1 + lane - lz;
                        }
                    }
                }
            }
            if (ballot(unknown)) refmask |= K1 >= W ? (~0ull << (K1 - W)) : ~0ull;
            for (uint32_t j = 0; j < W && j < 64; j++) { const uint32_t t = lane_get(tgt, j); if (t < 64) refmask |= 1ull << t; }
        }
        BVG_T1(15, tqp);
        const bool copied_from = (refmask >> lane) & 1ull;
        BVG_WC(12, __popcll(ballot(copied_from && on1)));                     // (work-count build: stored lists | reference-free with intervals | with reference and extras | direct)
        BVG_WC(13, __popcll(ballot(copied_from && on1 && ref == 0 && ic != 0)));
        BVG_WC(9, __popcll(ballot(copied_from && on1 && ref != 0 && (ic != 0 || nres != 0))));
        BVG_WC(14, __popcll(ballot(copied_from && on1 && ref == 0 && ic == 0)));
        const bool repn = on1 && x >= rep_lo && x < rep_hi;
        // MAT: a reported list WITHOUT reference is built in the pool like a stored one (it is decoded straight into place: the cheap path) and
        // copied out; a reported leaf WITH a reference (gl) is merged straight into the output by the position tasks
        const bool stored = copied_from || (MAT && repn && ref == 0 && d > 0);
        const bool gl = MAT && repn && !stored && d > 0;
        uint64_t gofs = 0;                                                    // where the node's successors go in a.succ
        if (MAT && repn) gofs = a.cum[x - a.from];
        // checksum key of the node (mix_node): a node outside [from, to) sums nothing (k1 = 0)
        uint32_t k0 = 0, k1 = 0;
        if (repn && !MAT) {
            node_key((uint64_t)x + a.node_base, k0, k1);
        }
        // a stored list without reference is emitted from its parked residuals (they play the referenced list): those are summed
        // there; every other residual is summed when it is decoded
        // ... unless it has no intervals either: then the list IS its residuals, decoded straight into its place (nothing parked, no level)
#ifdef BVG_NO_DIRECT
        const bool direct = false;
#else
        const bool direct = stored && ref == 0 && ic == 0;
#endif
        // ... and with intervals (28 % of the stored lists, 37 % of the positions the position tasks used to fill): the residuals still go
        // straight to their places -- each one shifted by the intervals that lie below it, which a task learns as it passes them -- and
        // whoever passes an interval records where it starts; the intervals themselves are filled in by the extras pass of level 0.
        // (not in the 85-VGPR instantiation: its state would spill, and sparse graphs hold few intervals; on1: `stored` is set for lanes
        // behind the super-row too when the peek is unknown)
        const bool d2 = D2 && OCC <= 5 && on1 && stored && ref == 0 && ic != 0;
        if (ballot(d2)) {                                                      // default start of interval k: behind every residual and the intervals before it
            if (d2) {
                uint32_t pre = nres;
                for (uint32_t kk = 0; kk < ic; kk++) { const uint32_t ln = (uint32_t)scr[ib + 2 * kk + 1] & 0xFFFFu; scr[ib + 2 * kk + 1] = (T)(ln | (pre << 16)); pre += ln; }
            }
            wave_sync();
        }
        const uint32_t k1d = (stored && ref == 0 && !direct && !d2) ? 0u : k1;
        // skip entries of the super-row, in node order
        const uint32_t cntE = (parse && nres >= kSkipMin) ? (nres - 1u) >> kSkipShift : 0u;
        uint32_t efirst;
        {
            const uint32_t eincl = wave_incl_scan32(cntE);
            efirst = sk_run + eincl - cntE;
            sk_run += lane_get(eincl, 63);
        }
        if (sk_run > sk_n) { failed = true; fail_need = 0xFFFFFFF5u; break; }   // index out of step with the stream
        BVG_T1(8, tq8);
        BVG_T1(5, tq5);

        // ================================================================== SUB-ROWS [sa, se) of the super-row
        cnt_super++; cnt_nodes += K1;
        uint32_t sa = 0;
        while (sa < K1) {
            const uint32_t tq8b = BVG_T0();
            const uint32_t avail = CAPe - pool_used;
            const bool cand = on1 && lane >= sa;
            const uint32_t size = (cand && stored) ? dclamp : 0u;
            const uint32_t rsz = (cand && (stored || gl) && !direct && !d2) ? (nres >= CAP ? CAP + 1 : nres + 1u) : 0u;     // parked residuals + the guard slot of the position tasks (MAT: also those of the leaves merged into the output)
            const uint32_t sincl = wave_incl_scan32(size), rincl = wave_incl_scan32(rsz);
            const bool fits = lane >= sa && lane < K1 && (uint64_t)sincl + rincl <= avail;
            const uint32_t se = sa + (uint32_t)__popcll(ballot(fits));          // (the sums are prefixes: `fits` is a contiguous run from sa)
            if (se == sa) {                                                   // the first node alone overflows the pool
                failed = true;
                uint32_t d0 = lane_get(d, sa); const uint32_t n0 = lane_get(nres, sa);
                if (d0 <= 0x3FFFFFFFu) d0 += (n0 > d0 ? d0 : n0) + 1u;
                fail_need = d0 > 0x3FFFFFFFu ? 0xFFFFFFF2u : d0 + pool_used + (d0 >> 2) + 64;
                break;
            }
            cnt_sub++;
            const bool act = cand && lane < se;
            const bool rep = act && repn;
            const uint32_t base = pool_used + (sincl - size);
            const uint32_t rtb = CAPe - (rincl > CAPe ? CAPe : rincl);
            if (act) nd_base[(uint32_t)x & RM] = (uint16_t)(stored ? base : kNoList);        // (a leaf has no list: nothing to compact, nothing to copy from)
            pool_used += lane_get(sincl, se - 1);
            BVG_T1(8, tq8b);
            const uint32_t tq9 = BVG_T0();
            // ---- D2: residuals (ResidualLongIterator, BVG:902-935): summed, and parked for the lists that are stored
            const bool rparse = parse && act;
            uint64_t csum = 0;
#if defined(BVG_PROF) && defined(BVG_PROF_WORK)
            (void)0;
#endif
            if (sk_n != 0 && ballot(rparse && cntE != 0)) {
                // long residual lists are cut at their skip entries: every segment of <= kSkipEvery gaps is one task.  Long tasks first
                // (full segments and tails of more than kShortTask gaps), the short tails after them: a pass of 64 tasks lasts as long
                // as its longest one, so like goes with like
                const bool hasres = rparse && nres > 0;
                const uint32_t ce = hasres ? cntE : 0u;
                const uint32_t lastc = nres - (ce << kSkipShift);
                const bool shortt = hasres && lastc <= kShortTask;
                const uint32_t Tn = hasres ? ce + (shortt ? 0u : 1u) : 0u;            // long tasks of this node
                const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, NL = lane_get(tincl, 63);
                const uint64_t smask = ballot(shortt);
                const uint32_t Ttot = NL + (uint32_t)__popcll(smask);
                const uint32_t slanes = smask ? ranked_lanes(smask) : 0u;             // lane k: the k-th lane with a short tail (wave-uniform test: every lane takes part)
                bool tbad = false;
                // RU tasks per lane and pass, decoded in one interleaved loop: two independent chains per lane hide each other's LDS latency
                auto task_passes = [&](auto RUc) {
                constexpr uint32_t RU = decltype(RUc)::value, RP = 64u * RU;
                for (uint32_t p0 = 0; p0 < Ttot; p0 += RP) {
                    BVG_WC(6, 1);
                    const uint32_t tq9p = BVG_T0();
                    bool tl[RU]; uint32_t cnt[RU], trel[RU], tpend[RU], tfirst[RU], taddr[RU], tk1[RU]; T r[RU];
                    uint32_t ivl[RU], ivn[RU], ivk[RU], ioff[RU], tic2[RU], tib2[RU], t0a[RU];   // lists decoded in place around their intervals (d2)
                    const bool anyd2 = D2 && OCC <= 5 && ballot(d2 && act) != 0;
#pragma unroll
                    for (uint32_t u = 0; u < RU; u++) {
                        const uint32_t t = p0 + 64u * u + lane;
                        tl[u] = t < Ttot;
                        // long tasks [0, NL): segment t - (tasks of the lanes before) of the lane that owns it; short tails [NL, Ttot): the
                        // last segment of the (t - NL)-th lane that has one
                        const bool isl = t < NL;
                        // (both searches are wave-uniformly skipped when this pass holds no task of their kind: the short tails sit in the last pass only)
                        const bool anylong = p0 + 64u * u < NL, anyshort = p0 + 64u * u + 63u >= NL && NL < Ttot;
                        const uint32_t lown = anylong ? task_owner(tincl, isl ? t : 0u) : 0u, sown = anyshort ? (uint32_t)__shfl((int)slanes, (int)((tl[u] && !isl) ? t - NL : 0u), 64) : 0u;
                        const int nl = tl[u] ? (int)(isl ? lown : sown) : (int)lane;
                        const uint32_t s_ts = (uint32_t)__shfl((int)ts, nl, 64);
                        const uint32_t t_rel = __shfl(rel, nl, 64), t_rec = __shfl(recrel, nl, 64), t_pend = __shfl(pend, nl, 64);
                        const uint32_t t_nres = __shfl(nres, nl, 64), t_dst = __shfl(stored ? ((direct || d2) ? base : rtb) : (gl ? rtb : kInf), nl, 64), t_ef = __shfl(efirst, nl, 64);
                        const uint32_t q = tl[u] ? (isl ? t - s_ts : (t_nres >= kSkipMin ? (t_nres - 1u) >> kSkipShift : 0u)) : 0u;     // a short tail is its node's last segment
                        const uint32_t s_k1 = __shfl(k1d, nl, 64);                 // (every lane takes part: a shuffle under a lane mask reads 0 from the masked lanes)
                        tk1[u] = tl[u] ? s_k1 : 0u;
                        const uint32_t t0 = q << kSkipShift;
                        const uint32_t t_ce = t_nres >= kSkipMin ? (t_nres - 1u) >> kSkipShift : 0u;      // the node's entries
                        cnt[u] = tl[u] ? (q == t_ce ? t_nres - t0 : kSkipEvery) : 0u;             // the last segment takes the remainder
                        trel[u] = tl[u] ? t_rel : 1u; r[u] = (T)((r0 - B) + nl); tpend[u] = tl[u] ? t_pend : kInf;   // (a lane without task reads at bit 1: win32p wants rel >= 1)
                        tfirst[u] = (tl[u] && q == 0) ? 1u : 0u;
                        taddr[u] = t_dst == kInf ? kInf : t_dst + t0;
                        if (tl[u] && q) {
                            const uint64_t ei = sk_base + t_ef + q - 1u;
                            trel[u] = t_rec + a.skip_bit[ei];
                            if (WIDE) {                                       // (the index of a wide graph holds 64-bit values)
                                const uint64_t sv = reinterpret_cast<const uint64_t*>(a.skip_val)[ei] - (uint64_t)B;
                                r[u] = (T)sv; if (sv >> 32) { tbad = true; cnt[u] = 0; }
                            } else r[u] = reinterpret_cast<const T*>(a.skip_val)[ei];
                            if (!(trel[u] > t_rel && trel[u] < t_pend)) { tbad = true; cnt[u] = 0; trel[u] = 1; }
                        }
                        ivl[u] = kInf; ivn[u] = 0; ivk[u] = 0; ioff[u] = 0; tic2[u] = 0; tib2[u] = 0; t0a[u] = t0;
                        if (anyd2) {                                          // (wave-uniform: the shuffles are executed by every lane)
                            const uint32_t s_d2 = (uint32_t)__shfl((int)(d2 ? 1 : 0), nl, 64), s_ic = __shfl(ic, nl, 64), s_ib = __shfl(ib, nl, 64);
                            if (tl[u] && s_d2 && cnt[u]) {
                                tic2[u] = s_ic; tib2[u] = s_ib;
                                if (q) while (ivk[u] < s_ic && scr[s_ib + 2 * ivk[u]] < r[u]) { ioff[u] += (uint32_t)scr[s_ib + 2 * ivk[u] + 1] & 0xFFFFu; ivk[u]++; }   // the intervals the tasks before this one have passed
                                if (ivk[u] < s_ic) { ivl[u] = (uint32_t)scr[s_ib + 2 * ivk[u]]; ivn[u] = (uint32_t)scr[s_ib + 2 * ivk[u] + 1] & 0xFFFFu; }
                            }
                        }
                    }
                    BVG_T1(14, tq9p);
                    if constexpr (Z3 && RU == 1) {
#include "bvg_scan_steps3.inc"
                    } else {
#include "bvg_scan_steps.inc"
                    }
                }
                };
#ifndef BVG_SCAN_RU2_FROM
#define BVG_SCAN_RU2_FROM 96
#endif
                // Two chains per lane only in the 128-VGPR instantiation: on the final structure the two-chain form measures flat there (thresholds 64 ... never: profiles/r04_ab_t0wait.txt),
                // and its registers are what the 85-VGPR instantiation of the sparse graphs spills (44 -> 31 spilled VGPRs without it: cnr-2000 +3.4 %, web +2.4 %, profiles/r04_ab_noru2*.txt)
                // (round 6: not for zeta_3 -- its one-chain step loop, bvg_scan_steps3.inc, issues 33 vector instructions per step against 63 per chain of the interleaved form)
#ifdef BVG_Z3_RU2
                constexpr bool ru2z3 = true;                                  // (A/B builds: the two-chain form for zeta_3 too)
#else
                constexpr bool ru2z3 = false;
#endif
                if ((!Z3 || ru2z3) && OCC == 4 && Ttot > (uint32_t)BVG_SCAN_RU2_FROM) task_passes(std::integral_constant<uint32_t, 2>{}); else task_passes(std::integral_constant<uint32_t, 1>{});
                bad |= tbad;
            } else if ((OCC == 6 || Z3) ? ballot(rparse && nres > 0) != 0 : false) {
                // no list of the sub-row is long enough for skip entries: one task per lane, the residuals of its own node, through the same branch-free
                // step loop.  Sparse graphs (the 85-VGPR instantiation) only: +1.0 % on cnr-2000, +0.7 % on `web`, 16 instead of 31 spilled registers
                // (profiles/r04_ab_lpn5_*.txt; as a lambda shared with the task passes +1.6 / +1.9 %, but then the dense instantiations lose 0.5 %:
                // r04_ab_lpn2_*.txt, r04_ab_lpn4_*.txt); the dense instantiations seldom come here and keep the plain loop below, their code unchanged
                constexpr uint32_t RU = 1;
                const bool has = rparse && nres > 0;
                const bool anyd2 = D2 && OCC <= 5 && ballot(d2 && act) != 0;
                bool tbad = false;
                uint32_t cnt[1] = {has ? nres : 0u}, trel[1] = {has ? rel : 1u}, tpend[1] = {has ? pend : kInf}, tfirst[1] = {1u}, tk1[1] = {has ? k1d : 0u};
                uint32_t taddr[1] = {(has && (stored || gl)) ? ((direct || d2) ? base : rtb) : kInf};
                T r[1] = {(T)(x - B)};
                uint32_t ivl[1] = {kInf}, ivn[1] = {0u}, ivk[1] = {0u}, ioff[1] = {0u}, tic2[1] = {0u}, tib2[1] = {0u}, t0a[1] = {0u};
                if (anyd2 && has && d2) { tic2[0] = ic; tib2[0] = ib; ivl[0] = (uint32_t)scr[ib]; ivn[0] = (uint32_t)scr[ib + 1] & 0xFFFFu; }
#ifndef BVG_ABLATE_LPN
                if constexpr (Z3) {
#include "bvg_scan_steps3.inc"
                } else {
#include "bvg_scan_steps.inc"
                }
#endif
                bad |= tbad;
            } else if (OCC != 6 && !Z3 && rparse) {
                if (nres > 0) {
                    T r = (T)(x - B);
                    uint32_t rr = rel;
                    uint32_t jk = 0, joff = 0, jl = kInf, jn = 0;                     // (d2: the next interval, what the passed ones add)
                    if (d2) { jl = (uint32_t)scr[ib]; jn = (uint32_t)scr[ib + 1] & 0xFFFFu; }
#ifdef BVG_ABLATE_LPN
                    if (false)
#endif
                    for (uint32_t t = 0; t < nres; t++) {
                        uint64_t val;
#ifdef BVG_EXP_DUMMY_LPN
                        { uint32_t dm = t; _Pragma("unroll") for (int z = 0; z < BVG_EXP_DUMMY_LPN; z++) asm volatile("v_xad_u32 %0, %0, %0, %0" : "+v"(dm)); if (dm == 0x12345u) err |= 1u; }
#endif
                        const uint32_t len = read_residual<false>(stage, rr, zfast, zk, a.cod.residual, val);
                        if (len == 0) { bad = true; break; }
                        rr += len;
                        if (WIDE) {
                            const int64_t tv = (int64_t)(uint64_t)r + (t == 0 ? nat2int64(val) : (int64_t)(1 + val));
                            if (((uint64_t)tv) >> 32) { bad = true; break; }
                        }
                        r = t == 0 ? (T)(r + (T)nat2int64(val)) : (T)(r + 1 + (T)val);
                        while (d2 && (uint32_t)r > jl) {
                            scr[ib + 2 * jk + 1] = (T)(jn | ((t + joff) << 16)); joff += jn; jk++;
                            if (jk < ic) { jl = (uint32_t)scr[ib + 2 * jk]; jn = (uint32_t)scr[ib + 2 * jk + 1] & 0xFFFFu; } else jl = kInf;
                        }
                        if (stored || gl) pool[((direct || d2) ? base : rtb) + t + joff] = r;
                        if (!MAT) csum += mix_node<T>(k1d, r);
                        if (rr > pend) { bad = true; break; }
                    }
                }
            }
            blk_chk += csum;
            if (ballot(bad)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
            wave_sync();
            BVG_T1(9, tq9);
            BVG_T1(5, tq8b);
            const uint32_t tq0 = BVG_T0();

            // ------------------------------------------------------------------ phase 2 of the sub-row
            constexpr uint32_t HS = 16;                                       // interval entry: length | position << HS
            const T HM = (T)0xFFFFu;
            uint32_t rlbN = 0, rlenS = rlenN;
            if (act && ref > 0) rlbN = nd_base[(uint32_t)(x - ref) & RM];
            // A stored node without reference is its residuals merged with its intervals (BVG:1087-1089): the parked residual values
            // play the role of an unmasked "referenced list", the intervals are the only extras to place (and the guard serves as its
            // empty array of residual positions).
            const bool pure = act && ref == 0;
            if (pure) { rlbN = rtb; rlenS = nres; }
            const uint32_t rtbN = pure ? rtb + nres : rtb, nresN = pure ? 0u : nres;
            const bool emitn = act && d > 0;
            // ---- level-synchronous emission by POSITION of the stored lists (as in bvg_rows.hip; no overlap checks: validated)
            const bool inrow = act && ref > 0 && ref + sa <= lane;                // the referenced list belongs to this sub-row
            uint32_t lvl = 0;
            for (int it = 0; it < 64; it++) {
                const uint32_t up = __shfl(lvl, inrow ? (int)(lane - ref) : (int)lane, 64);
                const uint32_t nl = inrow ? up + 1 : 0;
                const bool ch = nl != lvl; lvl = nl;
                if (!ballot(ch)) break;
            }
            const bool emits = emitn && (stored || gl) && !direct && !d2;
            const bool fills = act && d2;                                     // their intervals are written by the extras pass of level 0
            if (emits) pool[rtb + nres] = sentinel<T>();                      // guard behind the node's residual positions
            uint64_t remaining = ballot(emits || fills);
            wave_sync();
            BVG_T1(0, tq0);
            // WW: which of the lists with reference are built wave-wide
            const bool wwn = wwon && emits && ref > 0 && d <= kWWBits && rlenS <= kWWBits && d >= (a.dbg >> 16);   // (bits 16.. of BVG_DBG: the shortest list built this way)
            bool wwbad = false;
            uint64_t wsum = 0;
            for (uint32_t L = 0; remaining; L++) {
                const bool memall = emits && lvl == L, memf = fills && L == 0;
                const bool mem = memall && !wwn, memw = memall && wwn;
                remaining &= ~ballot(memall || memf);
                if (!ballot(memall || memf)) continue;
                BVG_WC(0, 1);
                if (ballot(mem || memf)) {
                // ZE: a bit vector per member (one bit per element of the list being built + a clear word behind it), all zero before the extras pass
                uint32_t ebase = 0; bool zel = false;
                if (zeon) {
                    const uint32_t ewn = mem ? ((d + 31u) >> 5) + 1u : 0u;
                    const uint32_t eincl = wave_incl_scan32(ewn), etot = lane_get(eincl, 63);
                    ebase = eincl - ewn;
                    zel = etot != 0 && etot <= kZEWords;                      // (does not fit: this level keeps the position-by-position tasks)
                    if (zel) { for (uint32_t w = lane; w < etot; w += 64) zem[w] = 0u; wave_sync(); }
                }
                // ---------------- Z1: one lane per extra: its output position = (extras below it) + (copied elements below it)
                const uint32_t tq3 = BVG_T0();
                {
                    const uint32_t In = mem ? nresN + ic : (memf ? ic : 0u);
                    const bool anyf = ballot(memf) != 0;
                    uint64_t fsum = 0;
                    const uint32_t iincl2 = wave_incl_scan32(In), is = iincl2 - In, Itot = lane_get(iincl2, 63);
#ifdef BVG_ABLATE_Z1
                    if (false)
#endif
                    for (uint32_t p0 = 0; p0 < Itot; p0 += 64) {
                        const bool tl = p0 + lane < Itot;
                        BVG_WC(3, 1);
                        const uint32_t own = deal_few(ballot(In != 0), iincl2, is, p0 + lane);
                        const int nl = tl ? (int)own : (int)lane;
                        const uint32_t s_first = (uint32_t)__shfl((int)is, nl, 64);
                        const uint32_t q = tl ? p0 + lane - s_first : 0u;
                        const uint32_t t_d = __shfl(d, nl, 64), t_rlb = __shfl(rlbN, nl, 64), t_rlen = __shfl(rlenS, nl, 64);
                        const uint32_t t_bc = __shfl(bc, nl, 64), t_sb = __shfl(sb, nl, 64), t_ic = __shfl(ic, nl, 64), t_ib = __shfl(ib, nl, 64);
                        const uint32_t t_nres = __shfl(nresN, nl, 64), t_rtb = __shfl(rtbN, nl, 64), t_ob = __shfl(base, nl, 64);
                        const T* const rl = pool + t_rlb; T* const rt = pool + t_rtb;
                        bool t_gl = false; int64_t* gp = nullptr;             // MAT: the task's list is a leaf merged straight into the output
                        if (MAT) {
                            const uint32_t s_gl = (uint32_t)__shfl((int)(gl ? 1 : 0), nl, 64), g_lo = __shfl((uint32_t)gofs, nl, 64), g_hi = __shfl((uint32_t)(gofs >> 32), nl, 64);
                            t_gl = s_gl != 0; gp = a.succ + (((uint64_t)g_hi << 32) | g_lo);
                        }
                        T vv = 0; uint32_t len = 1, pe = 0; bool isiv = false, isfill = false;
                        uint32_t f_k1 = 0;
                        uint32_t t_eb = 0;
                        if (anyf || zel) {                                    // (wave-uniform: the shuffles are executed by every lane)
                            const int s_f = __shfl((int)(memf ? 1 : 0), nl, 64);      // (hoisted: `tl && __shfl()` would run the shuffle under a lane mask)
                            isfill = tl && s_f != 0;
                            f_k1 = __shfl(k1, nl, 64);
                            if (zel) t_eb = __shfl(ebase, nl, 64);
                        }
                        if (isfill) {                                         // an interval of a list decoded in place: its start was recorded when the residuals passed it
                            vv = scr[t_ib + 2 * q]; const T pk = scr[t_ib + 2 * q + 1];
                            len = (uint32_t)(pk & HM); pe = (uint32_t)(pk >> HS);
                            if (pe + len > t_d) { pe = 0; len = 0; }
                        } else if (tl) {
                            uint32_t eb;
                            if (q < t_ic) {                                   // interval q: the intervals and residuals below it
                                isiv = true;
                                vv = scr[t_ib + 2 * q]; len = (uint32_t)(scr[t_ib + 2 * q + 1] & HM);
                                eb = 0;
                                for (uint32_t i = 0; i < q; i++) eb += (uint32_t)(scr[t_ib + 2 * i + 1] & HM);
                                eb += lds_lower_bound<T>(rt, t_nres, vv);
                            } else {                                          // residual q - ic
                                const uint32_t i = q - t_ic;
                                vv = rt[i]; eb = i;
                                for (uint32_t kk = 0; kk < t_ic; kk++) {
                                    const T leftv = scr[t_ib + 2 * kk]; const uint32_t ln = (uint32_t)(scr[t_ib + 2 * kk + 1] & HM);
                                    if (leftv <= vv) eb += ln;
                                }
                            }
                            uint32_t t = 0;
                            if (t_rlen) {                                     // copied elements below v: rank of its lower bound under the mask
                                const uint32_t qq = lds_lower_bound<T>(rl, t_rlen, vv);
                                uint32_t qn;
                                t = MaskPrefix<T>::rank(scr + t_sb, t_bc, t_rlen, qq, qn);
                            }
                            pe = eb + t;
                            if (pe + len > t_d) { pe = 0; len = 0; }          // (cannot happen in a validated block; never write outside the list)
                        }
                        wave_sync();                                      // the parked values have been read: positions may replace them
                        if (tl && len) {
                            if (isfill) { for (uint32_t i = 0; i < len; i++) { pool[t_ob + pe + i] = (T)(vv + i); if (!MAT) fsum += mix_node<T>(f_k1, (T)(vv + i)); } }
                            else if (isiv && zel) {                           // ZE: the interval is written here, element by element (LongIntervalSequenceIterator.java:71-78), and its places are marked
                                for (uint32_t i = 0; i < len; i++) { pool[t_ob + pe + i] = (T)(vv + i); fsum += mix_node<T>(f_k1, (T)(vv + i)); atomicOr(&zem[t_eb + ((pe + i) >> 5)], 1u << ((pe + i) & 31u)); }
                            }
                            else if (isiv) scr[t_ib + 2 * q + 1] = (T)len | (T)((T)pe << HS);
                            else {
                                if (MAT && t_gl) gp[pe] = (int64_t)((uint64_t)vv + nbase); else pool[t_ob + pe] = vv;
                                rt[q - t_ic] = (T)pe;
                                if (zel) atomicOr(&zem[t_eb + (pe >> 5)], 1u << (pe & 31u));
                            }
                        }
                        wave_sync();
                    }
                    blk_chk += fsum;
                }
                BVG_T1(3, tq3);
                const uint32_t tq1 = BVG_T0();
                if (zel) {
                // ---------------- Z2 by KEPT element (ZE): tasks of S consecutive kept elements of a member's referenced list (MaskedLongIterator.java:73-100), each written to
                // the next clear bit of the member's bit vector -- the residuals and intervals are in place already (and summed), so no step is spent on their positions
                const uint32_t ncl = mem ? (pure ? nres : ncopN) : 0u;
                const uint32_t Wc = wave_sum32(ncl), Nc = (uint32_t)__popcll(ballot(ncl != 0));
                uint32_t S = Nc < 64u ? (Wc + (63u - Nc)) / (64u - Nc) : 0x7FFFFFFFu;
                if (S < kScanMinTask) S = kScanMinTask;
                uint32_t Tn = 0;
                if (ncl) { Tn = (uint32_t)((float)ncl / (float)S); while (Tn * S < ncl) Tn++; while (Tn > 1u && (Tn - 1u) * S >= ncl) Tn--; }
                const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, Ttot = lane_get(tincl, 63);
                uint64_t zsum = 0;
                for (uint32_t p0 = 0; p0 < Ttot; p0 += 64) {
                    const bool tl = p0 + lane < Ttot;
                    BVG_WC(1, 1); BVG_WCL(2, tl ? 1u : 0u);
                    const uint32_t own = deal_few(ballot(Tn != 0), tincl, ts, p0 + lane);
                    const int nl = tl ? (int)own : (int)lane;
                    const uint32_t s_first = (uint32_t)__shfl((int)ts, nl, 64);
                    const uint32_t q = tl ? p0 + lane - s_first : 0u;
                    const uint32_t t_nc = __shfl(ncl, nl, 64), t_rlb = __shfl(rlbN, nl, 64), t_rlen = __shfl(rlenS, nl, 64);
                    const uint32_t t_bc = __shfl(bc, nl, 64), t_sb = __shfl(sb, nl, 64), t_ob = __shfl(base, nl, 64), t_k1 = __shfl(k1, nl, 64), t_eb = __shfl(ebase, nl, 64);
                    const T* const rl = pool + t_rlb; T* const out = pool + t_ob; const uint32_t* const ew = zem + t_eb; const T* const blkp = scr + t_sb;
                    uint32_t t = 0, tstop = 0, qcur = 0, krem = kInf, bi = t_bc, p = 0;
                    if (tl) {
                        t = q * S; tstop = t + S < t_nc ? t + S : t_nc;
                        if (t_rlen) MaskPrefix<T>::select(blkp, t_bc, t_rlen, t, qcur, krem, bi);   // the t-th kept position of the referenced list
                        // the t-th clear bit of the member's bit vector: whole words first, then inside the word
                        uint32_t k = t, wi = 0, inv = ~ew[0];
                        for (;;) { const uint32_t z = (uint32_t)__builtin_popcount(inv); if (k < z) break; k -= z; wi++; inv = ~ew[wi]; }   // (ends: the word behind the list is clear)
                        uint32_t pos = 0;
#pragma unroll
                        for (uint32_t st = 16; st; st >>= 1) { const uint32_t c = (uint32_t)__builtin_popcount(inv & ((1u << (pos + st)) - 1u)); if (c <= k) pos += st; }
                        p = (wi << 5) + pos;
                    }
                    const uint32_t rlast = t_rlen ? t_rlen - 1u : 0u;
                    BVG_WCL(5, tstop - t); BVG_WC(4, wave_max32(tstop - t));
                    for (; t < tstop; t++) {
                        const T v = rl[qcur < rlast ? qcur : rlast];
                        const T e0 = blkp[bi], e1 = blkp[bi + 1u];                        // (reads past the node's blocks stay inside the scratch area / the window)
                        out[p] = v;
                        zsum += mix_node<T>(t_k1, v);
                        qcur++; krem--;
                        const bool cross = krem == 0;                                     // the keep block ended: skip block bi, enter keep block bi + 1
                        const uint32_t p0e = MaskPrefix<T>::pos(e0);
                        const uint32_t nq = bi < t_bc ? p0e : t_rlen;
                        const uint32_t nk = bi + 1u < t_bc ? MaskPrefix<T>::pos(e1) - p0e : kInf;
                        qcur = cross ? nq : qcur; krem = cross ? nk : krem; bi += cross ? 2u : 0u;
                        uint32_t pn = p + 1u;                                             // the next clear bit behind p
                        uint32_t w = ~ew[pn >> 5] >> (pn & 31u);
                        while (w == 0u) { pn = (pn | 31u) + 1u; w = ~ew[pn >> 5]; }
                        p = pn + (uint32_t)__builtin_ctz(w);
                    }
                    wave_sync();
                }
                blk_chk += zsum;
                } else {
                // ---------------- Z2: tasks of S output positions, all equally long
                // S in one step: sum_i ceil(d_i / S) <= W / S + N - N / S < 64 once S >= W / (64 - N)  (N lists, W positions in all)
                const uint32_t Wl = wave_sum32(mem ? d : 0u), Nl = (uint32_t)__popcll(ballot(mem));
#ifdef BVG_OLD_S
                uint32_t S = (Wl + 63u) >> 6; if (S < kScanMinTask) S = kScanMinTask;
                uint32_t Tn = 0;
                for (int it = 0; it < 6; it++) {
                    Tn = 0;
                    if (mem) { Tn = (uint32_t)((float)d / (float)S); while (Tn * S < d) Tn++; while (Tn > 1u && (Tn - 1u) * S >= d) Tn--; }
                    const uint32_t tt = wave_sum32(Tn);
                    if (tt <= 64u || it == 5) break;
                    const uint32_t s2 = (uint32_t)((float)S * (float)tt * (1.0f / 64.0f));
                    S = s2 > S ? s2 : S + 1u;
                }
#else
                uint32_t S = Nl < 64u ? (Wl + (63u - Nl)) / (64u - Nl) : 0x7FFFFFFFu;
                if (S < kScanMinTask) S = kScanMinTask;
                uint32_t Tn = 0;
                if (mem) { Tn = (uint32_t)((float)d / (float)S); while (Tn * S < d) Tn++; while (Tn > 1u && (Tn - 1u) * S >= d) Tn--; }
#endif
                const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, Ttot = lane_get(tincl, 63);
                BVG_T1(1, tq1);
#ifdef BVG_ABLATE_Z2
                if (false)
#endif
                for (uint32_t p0 = 0; p0 < Ttot; p0 += 64) {
                    const uint32_t tq2 = BVG_T0();
                    const bool tl = p0 + lane < Ttot;                     // task of this lane: (node lane, task index inside the node)
                    BVG_WC(1, 1); BVG_WCL(2, tl ? 1u : 0u);
                    const uint32_t own = deal_few(ballot(Tn != 0), tincl, ts, p0 + lane);
                    const int nl = tl ? (int)own : (int)lane;
                    const uint32_t s_first = (uint32_t)__shfl((int)ts, nl, 64);
                    const uint32_t q = tl ? p0 + lane - s_first : 0u;
                    const uint32_t t_d = __shfl(d, nl, 64), t_rlb = __shfl(rlbN, nl, 64), t_rlen = __shfl(rlenS, nl, 64);
                    const uint32_t t_bc = __shfl(bc, nl, 64), t_sb = __shfl(sb, nl, 64), t_ic = __shfl(ic, nl, 64), t_ib = __shfl(ib, nl, 64);
                    const uint32_t t_nres = __shfl(nresN, nl, 64), t_rtb = __shfl(rtbN, nl, 64), t_ob = __shfl(base, nl, 64);
                    const uint32_t t_k1 = __shfl(k1, nl, 64);
                    const T* const rl = pool + t_rlb; const T* const rt = pool + t_rtb; T* const out = pool + t_ob;
                    bool t_gl = false; int64_t* gp = nullptr;
                    if (MAT) {
                        const uint32_t s_gl = (uint32_t)__shfl((int)(gl ? 1 : 0), nl, 64), g_lo = __shfl((uint32_t)gofs, nl, 64), g_hi = __shfl((uint32_t)(gofs >> 32), nl, 64);
                        t_gl = s_gl != 0; gp = a.succ + (((uint64_t)g_hi << 32) | g_lo);
                    }
                    uint32_t p = 0, pstop = 0, ri = 0, rnext = kInf, ivk = t_ic, ivpos = kInf, ivlen = 0, qcur = 0, krem = kInf, bi = t_bc;
                    T ivleft = 0;
                    if (tl) {
                        p = q * S; pstop = p + S < t_d ? p + S : t_d;
                        ri = lds_lower_bound<T>(rt, t_nres, (T)p);            // residual positions below p
                        rnext = (uint32_t)rt[ri];                             // (the guard reads as kInf)
                        uint32_t ie = 0;
                        for (uint32_t i = 0; i < t_ic; i++) {                 // interval elements below p; the interval at / after p
                            const T pk = scr[t_ib + 2 * i + 1];
                            const uint32_t ln = (uint32_t)(pk & HM), ps = (uint32_t)(pk >> HS);
                            if (ps + ln > p) { ivk = i; ivpos = ps; ivlen = ln; ivleft = scr[t_ib + 2 * i]; if (p > ps) ie += p - ps; break; }
                            ie += ln;
                        }
                        const uint32_t t = p - ri - ie;                       // rank of the next copied element among the kept ones
                        if (t_rlen) MaskPrefix<T>::select(scr + t_sb, t_bc, t_rlen, t, qcur, krem, bi);   // MaskedLongIterator.java:73-100: the t-th kept position
                    }
                    const uint32_t rlast = t_rlen ? t_rlen - 1u : 0u;
                    uint64_t zsum = 0;
                    BVG_WCL(5, pstop - p);
                    BVG_T1(2, tq2);
                    const uint32_t tq4 = BVG_T0();
                    // One output position per step and lane.  With some fifty lanes at work nearly every step sees a lane at the end of a
                    // copy block and another at a residual, so those two are handled without branches (the next block's entries and the
                    // next residual position are read in every step, and selected); only the end of an interval -- rarer -- is a branch.
                    const T* const blkp = scr + t_sb;
                    for (;;) {
#ifdef BVG_ABLATE_Z2LOOP
                        break;
#endif
                        const bool todo = p < pstop;
                        if (!ballot(todo)) break;
                        BVG_WC(4, 1);
#ifdef BVG_EXP_DUMMY
                        { uint32_t dm = p; _Pragma("unroll") for (int z = 0; z < BVG_EXP_DUMMY; z++) asm volatile("v_xad_u32 %0, %0, %0, %0" : "+v"(dm)); if (dm == 0x12345u) err |= 1u; }
#endif
                        const T cv = rl[qcur < rlast ? qcur : rlast];
                        const T e0 = blkp[bi], e1 = blkp[bi + 1u];                        // (reads past the node's blocks stay inside the scratch area / the window)
                        const bool isr = p == rnext;                                      // a residual: placed by Z1 (and summed when it was decoded)
                        const uint32_t io = p - ivpos;
                        const bool ii = io < ivlen;                                       // LongIntervalSequenceIterator.java:71-78
                        const bool emit = todo && !isr;
                        const T vv = ii ? (T)(ivleft + (T)io) : cv;
                        if (MAT) { if (emit) { if (t_gl) gp[p] = (int64_t)((uint64_t)vv + nbase); else out[p] = vv; } }
                        else {
                            if (emit) out[p] = vv;
                            zsum += mix_node<T>(emit ? t_k1 : 0u, vv);
                        }
                        const bool cp = emit && !ii;                                      // a copied element: MaskedLongIterator.java:81-100
                        qcur += cp ? 1u : 0u; krem -= cp ? 1u : 0u;
                        const bool cross = cp && krem == 0;                               // the keep block ended: skip block bi, enter keep block bi + 1
                        const uint32_t p0e = MaskPrefix<T>::pos(e0);
                        const uint32_t nq = bi < t_bc ? p0e : t_rlen;
                        const uint32_t nk = bi + 1u < t_bc ? MaskPrefix<T>::pos(e1) - p0e : kInf;
                        qcur = cross ? nq : qcur; krem = cross ? nk : krem; bi += cross ? 2u : 0u;
                        ri += (todo && isr) ? 1u : 0u;
                        rnext = (uint32_t)rt[ri];                                         // (the guard reads as kInf)
                        if (emit && ii && io + 1u == ivlen) {
                            ivk++; ivpos = kInf; ivlen = 0;
                            if (ivk < t_ic) { const T pk = scr[t_ib + 2 * ivk + 1]; ivlen = (uint32_t)(pk & HM); ivpos = (uint32_t)(pk >> HS); ivleft = scr[t_ib + 2 * ivk]; }
                        }
                        p += todo ? 1u : 0u;
                    }
                    blk_chk += zsum;
                    wave_sync();
                    BVG_T1(4, tq4);
                }
                }   // (Z2 by position)
                }   // (the position tasks: members that are not built wave-wide)
                if (WWT && wwon) {
                // ---------------- WW: the members of this level that are built WAVE-WIDE, one list after the other (their referenced lists are complete: they belong to
                // an earlier level, sub-row or super-row).  Everything about a list is wave-uniform (read off its lane into scalar registers); the 64 lanes are 64
                // consecutive positions.  (1) The copy mask (MaskedLongIterator.java:73-100) as a bit vector over the referenced list: every block end toggles one bit
                // (lanes = blocks, ds_xor), an inclusive prefix-XOR of a 64-bit word on the scalar unit is the block-index parity of its 64 positions, kept = even; the
                // kept elements are compacted to the FRONT of the list being built (rank = v_mbcnt under the mask).  (2) Extras: a lane per residual / interval finds
                // how many kept elements lie below it (lower bound in the compacted front) -- its output position (MergedLongIterator.java:54-92 on disjoint streams)
                // -- and sets that bit in the second bit vector.  (3) The kept elements are spread IN PLACE from the last 64 positions down to the first: position o
                // holds front[o - (extras below o)] unless its bit is set; a set bit is the next residual (or an interval element, filled behind).  One multiply-add
                // per copied / interval element; the residuals were summed when they were decoded.  No binary search per output position, no dealing, no levels inside.
                uint64_t wm = ballot(memw);
                while (wm) {
                    const uint32_t j = (uint32_t)__ffsll((unsigned long long)wm) - 1u; wm &= wm - 1ull;
                    const uint32_t n_d = lane_get(d, j), n_rlb = lane_get(rlbN, j), n_rlen = lane_get(rlenS, j), n_bc = lane_get(bc, j), n_sb = lane_get(sb, j);
                    const uint32_t n_ic = lane_get(ic, j), n_ib = lane_get(ib, j), n_nres = lane_get(nresN, j), n_rtb = lane_get(rtbN, j), n_ob = lane_get(base, j), n_k1 = lane_get(k1, j);
                    const T* const rl = pool + n_rlb; const T* const rt = pool + n_rtb; T* const out = pool + n_ob;
                    // (1) copy mask -> bit vector -> compaction
                    const uint32_t tw = (n_rlen + 63u) >> 6;
                    if (lane < 2u * tw) wwm[lane] = 0u;
                    wave_sync();
                    for (uint32_t b0 = 0; b0 < n_bc; b0 += 64) {
                        const uint32_t bi = b0 + lane;
                        if (bi < n_bc) { const uint32_t eb = MaskPrefix<T>::pos(scr[n_sb + bi]); if (eb < n_rlen) atomicXor(&wwm[eb >> 5], 1u << (eb & 31u)); }
                    }
                    wave_sync();
                    uint32_t kept = 0, carry = 0;
                    for (uint32_t c = 0; c < tw; c++) {
                        const uint32_t tlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)wwm[2u * c]), thi = (uint32_t)__builtin_amdgcn_readfirstlane((int)wwm[2u * c + 1u]);
                        const uint64_t tg = ((uint64_t)thi << 32) | tlo;
                        uint64_t x = tg; x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;    // bit p: parity of the block ends at or below position 64c + p
                        if (carry) x = ~x;
                        carry ^= (uint32_t)__popcll(tg) & 1u;
                        uint64_t km = ~x;                                     // even block index: kept (behind the last block: kept iff their number is even -- the same parity)
                        const uint32_t rem = n_rlen - 64u * c;
                        if (rem < 64u) km &= (1ull << rem) - 1ull;
                        if (__builtin_amdgcn_inverse_ballot_w64(km)) {
                            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                            out[kept + rk] = rl[64u * c + lane];
                        }
                        kept += (uint32_t)__popcll(km);
                    }
#ifdef BVG_WW_TRACE
                    if (lane == 0) printf("WW node %lld d %u rlen %u bc %u kept %u nres %u ic %u\n", (long long)(r0 + j), n_d, n_rlen, n_bc, kept, n_nres, n_ic);
#endif
                    if (kept > n_d) { wwbad = true; break; }                  // (cannot happen in a validated block)
                    // (2) the extras' positions -> second bit vector
                    const uint32_t ew = (n_d + 63u) >> 6;
                    wave_sync();
                    if (lane < 2u * ew) wwm[lane] = 0u;
                    wave_sync();
                    const uint32_t fstep = kept ? 1u << (31u - (uint32_t)__builtin_clz(kept)) : 0u;      // branch-free lower bound in the front: the largest power of two <= kept
                    auto front_lb = [&](T v) {                                // elements of the compacted front below v
                        uint32_t lo = 0;
                        for (uint32_t st = fstep; st; st >>= 1) { const uint32_t m = lo + st; if (m <= kept && out[m - 1u] < v) lo = m; }
                        return lo;
                    };
                    if (n_ic) {
                        const uint32_t rstep = n_nres ? 1u << (31u - (uint32_t)__builtin_clz(n_nres)) : 0u;
                        for (uint32_t q0 = 0; q0 < n_ic; q0 += 64) {
                            const uint32_t q = q0 + lane;
                            if (q < n_ic) {
                                const T leftv = scr[n_ib + 2u * q]; const uint32_t ln = (uint32_t)(scr[n_ib + 2u * q + 1u] & HM);
                                uint32_t eb = 0;
                                for (uint32_t k = 0; k < q; k++) eb += (uint32_t)(scr[n_ib + 2u * k + 1u] & HM);
                                uint32_t lo = 0;
                                for (uint32_t st = rstep; st; st >>= 1) { const uint32_t m = lo + st; if (m <= n_nres && rt[m - 1u] < leftv) lo = m; }
                                const uint32_t pe = eb + lo + front_lb(leftv);
                                if (pe + ln <= n_d) {
                                    for (uint32_t i = 0; i < ln; i++) atomicOr(&wwm[(pe + i) >> 5], 1u << ((pe + i) & 31u));
                                    scr[n_ib + 2u * q + 1u] = (T)ln | (T)((T)pe << HS);
                                } else scr[n_ib + 2u * q + 1u] = (T)0;        // (cannot happen in a validated block; never write outside the list)
                            }
                        }
                    }
                    for (uint32_t i0 = 0; i0 < n_nres; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        if (i < n_nres) {
                            const T e = rt[i];
                            uint32_t pe = i + front_lb(e);
                            for (uint32_t k = 0; k < n_ic; k++) { if (scr[n_ib + 2u * k] <= e) pe += (uint32_t)(scr[n_ib + 2u * k + 1u] & HM); }
                            if (pe < n_d) atomicOr(&wwm[pe >> 5], 1u << (pe & 31u));
                        }
                    }
                    wave_sync();
                    // (3) spread the front over the list, from the top down
                    uint32_t eabove = n_d - kept;                             // extras at or above the current chunk's first position
                    for (uint32_t c = ew; c-- > 0u;) {
                        const uint32_t elo = (uint32_t)__builtin_amdgcn_readfirstlane((int)wwm[2u * c]), ehi = (uint32_t)__builtin_amdgcn_readfirstlane((int)wwm[2u * c + 1u]);
                        const uint64_t em = ((uint64_t)ehi << 32) | elo;
                        const uint32_t ebelow = eabove - (uint32_t)__popcll(em);
                        const uint32_t o = 64u * c + lane;
                        const bool valid = o < n_d, isE = __builtin_amdgcn_inverse_ballot_w64(em);
                        const uint32_t re = ebelow + __builtin_amdgcn_mbcnt_hi(ehi, __builtin_amdgcn_mbcnt_lo(elo, 0u));     // extras below position o
                        const bool cpy = valid && !isE && re <= o;
                        T v = 0;
                        if (cpy) v = out[o - re];
                        uint32_t ridx = re; bool isiv = false;
                        for (uint32_t k = 0; k < n_ic; k++) {                 // (wave-uniform: the list's intervals)
                            const T pk = scr[n_ib + 2u * k + 1u];
                            const uint32_t ps = (uint32_t)(pk >> HS), ln = (uint32_t)(pk & HM);
                            const uint32_t dd = o - ps;
                            isiv = isiv || dd < ln;
                            ridx -= o > ps ? (dd < ln ? dd : ln) : 0u;
                        }
                        const bool isR = valid && isE && !isiv && ridx < n_nres;
                        T rv = 0;
                        if (isR) rv = rt[ridx];
                        wave_sync();                                          // every element of the chunk is read before any is written
                        if (cpy) { out[o] = v; wsum += mix_node<T>(n_k1, v); }
                        if (isR) out[o] = rv;
                        wave_sync();
                        eabove = ebelow;
                    }
                    // (4) the intervals (LongIntervalSequenceIterator.java:57-78): a lane each
                    for (uint32_t q0 = 0; q0 < n_ic; q0 += 64) {
                        const uint32_t q = q0 + lane;
                        if (q < n_ic) {
                            const T leftv = scr[n_ib + 2u * q]; const T pk = scr[n_ib + 2u * q + 1u];
                            const uint32_t ps = (uint32_t)(pk >> HS), ln = (uint32_t)(pk & HM);
                            for (uint32_t i = 0; i < ln; i++) { out[ps + i] = (T)(leftv + i); wsum += mix_node<T>(n_k1, (T)(leftv + i)); }
                        }
                    }
                    wave_sync();
                }
                }
            }
            blk_chk += wsum;
            if (ballot(wwbad)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
            // ---------------- leaf pass: the run queue, cut into chunks of kChunk elements dealt to all lanes.  Every referenced list
            // is complete by now.  A chunk is straight-line work: 4 elements per step, their LDS reads issued together.
            const uint32_t tqL = BVG_T0();
            // a leaf's items: its kept copy blocks -- block 2j of the copy mask, and the implicit tail behind an even number of blocks
            // (MaskedLongIterator.java:73-78): runs [start, end) of the referenced list -- and its intervals
            const bool leaf = !MAT && emitn && !stored && rep;
            const uint32_t nkept = (leaf && ref > 0) ? ((bc + 2u) >> 1) : 0u;
            const uint32_t Ln = leaf ? nkept + ic : 0u;
            const uint32_t lincl = wave_incl_scan32(Ln), lfirst = lincl - Ln, Q = lane_get(lincl, 63);
#ifdef BVG_ABLATE_LEAF
            if (false)
#endif
            for (uint32_t d0 = 0; d0 < Q; d0 += 64) {
                const bool dl = d0 + lane < Q;
                BVG_WC(15, 1);
                const uint32_t iown = task_owner(lincl, d0 + lane);
                const int il = dl ? (int)iown : (int)lane;
                const uint32_t i_first = (uint32_t)__shfl((int)lfirst, il, 64), i_nk = (uint32_t)__shfl((int)nkept, il, 64), i_bc = (uint32_t)__shfl((int)bc, il, 64);
                const uint32_t i_sb = (uint32_t)__shfl((int)sb, il, 64), i_ib = (uint32_t)__shfl((int)ib, il, 64);
                const uint32_t i_rlb = (uint32_t)__shfl((int)rlbN, il, 64), i_rlen = (uint32_t)__shfl((int)rlenS, il, 64);
                uint32_t e_lo = 0, e_hi = 0;
                if (dl) {
                    const uint32_t j = d0 + lane - i_first;
                    if (j < i_nk) {
                        const uint32_t bi = 2u * j;
                        const uint32_t st0 = bi ? MaskPrefix<T>::pos(scr[i_sb + bi - 1u]) : 0u;
                        const uint32_t en0 = bi < i_bc ? MaskPrefix<T>::pos(scr[i_sb + bi]) : i_rlen;
                        e_lo = i_rlb + st0; e_hi = (en0 > st0 ? en0 - st0 : 0u) | ((uint32_t)il << 24);
                    } else {
                        const uint32_t k = j - i_nk;
                        e_lo = (uint32_t)scr[i_ib + 2u * k]; e_hi = (uint32_t)scr[i_ib + 2u * k + 1u] | ((uint32_t)il << 24) | 0x80000000u;
                    }
                }
                const uint32_t e_len = e_hi & 0xFFFFFFu;
                const uint32_t nch = (e_len + kChunk - 1u) / kChunk;
                const uint32_t cincl = wave_incl_scan32(nch), cs = cincl - nch, Ctot = lane_get(cincl, 63);
                const uint32_t e_k1 = __shfl(k1, (int)((e_hi >> 24) & 63u), 64);
                for (uint32_t p0 = 0; p0 < Ctot; p0 += 64) {
                    const bool tl = p0 + lane < Ctot;
                    const uint32_t own = task_owner(cincl, p0 + lane);           // (every lane takes part in the shuffles)
                    const int sl = tl ? (int)own : (int)lane;
                    const uint32_t s_first = (uint32_t)__shfl((int)cs, sl, 64);
                    const uint32_t q = tl ? p0 + lane - s_first : 0u;
                    const uint32_t c_lo = __shfl(e_lo, sl, 64), c_hi = __shfl(e_hi, sl, 64);
                    const uint32_t c_k1 = __shfl(e_k1, sl, 64);
                    const uint32_t c_len = c_hi & 0xFFFFFFu, o = q * kChunk;
                    const uint32_t n = tl ? (c_len - o < kChunk ? c_len - o : kChunk) : 0u;
                    const bool iota = (c_hi >> 31) != 0u;
                    const uint32_t b0 = c_lo + o;                              // first pool element / first value of the chunk
                    const T* const src = pool + (iota ? 0u : b0);
                    uint64_t lsum = 0;
                    const uint32_t nmax = wave_max32(n);
                    BVG_WC(10, 1); BVG_WC(11, (nmax + 3u) >> 2);
                    const uint32_t tqL2 = BVG_T0();
#ifdef BVG_ABLATE_LEAFLOOP
                    if (false)
#endif
                    for (uint32_t i = 0; i < nmax; i += 4) {
                        const T v0 = src[i], v1 = src[i + 1], v2 = src[i + 2], v3 = src[i + 3];   // (reads past a run stay inside the LDS allocation)
                        lsum += mix_node<T>(i < n ? c_k1 : 0u, iota ? (T)(b0 + i) : v0);
                        lsum += mix_node<T>(i + 1 < n ? c_k1 : 0u, iota ? (T)(b0 + i + 1) : v1);
                        lsum += mix_node<T>(i + 2 < n ? c_k1 : 0u, iota ? (T)(b0 + i + 2) : v2);
                        lsum += mix_node<T>(i + 3 < n ? c_k1 : 0u, iota ? (T)(b0 + i + 3) : v3);
                    }
                    blk_chk += lsum;
                    BVG_T1(11, tqL2);
                    wave_sync();
                }
            }
            BVG_T1(10, tqL);
            if (rep) { blk_arcs += d; blk_nodes += 1; if (!MAT) blk_chk += mix_node_const(k0, k1, nbase, d); }
            if (MAT) {
                // the lists of the sub-row that were built in the pool leave in coalesced runs (one list after the other: a wave-uniform loop)
                wave_sync();
                uint64_t cm = ballot(rep && stored && d > 0);
                while (cm) {
                    const uint32_t j = (uint32_t)__ffsll((unsigned long long)cm) - 1u; cm &= cm - 1ull;
                    const uint32_t sb0 = lane_get(base, j), dd = lane_get(d, j);
                    int64_t* const dst = a.succ + lane_get64(gofs, j);
                    for (uint32_t t = lane; t < dd; t += 64) dst[t] = (int64_t)((uint64_t)pool[sb0 + t] + nbase);
                }
                if (rep && a.outdeg) a.outdeg[x - a.from] = (int32_t)d;
                if (act && stored && !copied_from) nd_base[(uint32_t)x & RM] = (uint16_t)kNoList;   // nobody copies from it: the next compaction drops it
            }
            wave_sync();
            sa = se;
            if (sa < K1) compact(r0 + (int64_t)sa);                            // the next sub-row starts from the stored lists of the W nodes before it
        }
        if (failed) break;
        // next super-row: reuse the prefetched offsets
        r0 += K1;
        off_x = nxt_off; rec_end = nxt_end;
    }

    err = wave_or32(err);
    if (failed) {
        if (lane == 0) {
            uint32_t slot = atomicAdd(a.fail_count, 1u);
            if (slot < a.fail_cap) { a.fail_list[slot] = bid; if (a.fail_need) a.fail_need[slot] = fail_need; }
        }
        return;
    }
    blk_arcs = wave_sum64(blk_arcs); blk_chk = wave_sum64(blk_chk); blk_nodes = wave_sum64(blk_nodes);
    if (lane == 0) {
        unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;   // this block's result stripe
        atomicAdd(&accs[0], (unsigned long long)blk_arcs);
        atomicAdd(&accs[1], (unsigned long long)blk_chk);
        atomicAdd(&accs[2], (unsigned long long)blk_nodes);
        if (err) atomicOr(&accs[3], (unsigned long long)err);
        if (a.dbg & 64u) { atomicAdd(&a.acc[5], (unsigned long long)cnt_super); atomicAdd(&a.acc[6], (unsigned long long)cnt_sub); atomicAdd(&a.acc[7], (unsigned long long)cnt_nodes); }
#ifdef BVG_PROF
        if (a.dbg & 64u) {
            for (int i = 0; i < 10; i++) atomicAdd(&a.acc[9 + i], (unsigned long long)cyc[i]);
            atomicAdd(&a.acc[20], (unsigned long long)cyc[10]); atomicAdd(&a.acc[21], (unsigned long long)cyc[11]);
            for (int i = 12; i < 16; i++) atomicAdd(&a.acc[12 + i], (unsigned long long)cyc[i]);
#ifdef BVG_PROF_WORK
            for (int i = 0; i < 4; i++) atomicAdd(&a.acc[28 + i], (unsigned long long)cyc2[i]);
#endif
        }
#endif
    }
}

}  // namespace

// what the kernel needs of LDS besides the pool and the scratch area (static arrays)
size_t scan_static_lds() { return (size_t)kRing * 4; }   // (the node ring: part of the dynamic allocation since round 6, behind the window; the host's footprint sum is unchanged)

template <int OCC, bool D2> static void launch_scan_occ(const DecodeArgs& a, uint32_t nblocks, bool wide, size_t dyn, hipStream_t s) {
    const bool z3 = a.cod.zeta_k == 3;
    if (wide) { if (z3) hipLaunchKernelGGL((scan_kernel<true, true, OCC, D2, false>), dim3(nblocks), dim3(64), dyn, s, a); else hipLaunchKernelGGL((scan_kernel<false, true, OCC, D2, false>), dim3(nblocks), dim3(64), dyn, s, a); }
    else { if (z3) hipLaunchKernelGGL((scan_kernel<true, false, OCC, D2, false>), dim3(nblocks), dim3(64), dyn, s, a); else hipLaunchKernelGGL((scan_kernel<false, false, OCC, D2, false>), dim3(nblocks), dim3(64), dyn, s, a); }
}
// the materialising form: no checksum
template <bool D2> static void launch_scan_mat(const DecodeArgs& a, uint32_t nblocks, bool wide, size_t dyn, hipStream_t s) {
    const bool z3 = a.cod.zeta_k == 3;
    if (wide) { if (z3) hipLaunchKernelGGL((scan_kernel<true, true, 4, D2, true>), dim3(nblocks), dim3(64), dyn, s, a); else hipLaunchKernelGGL((scan_kernel<false, true, 4, D2, true>), dim3(nblocks), dim3(64), dyn, s, a); }
    else { if (z3) hipLaunchKernelGGL((scan_kernel<true, false, 4, D2, true>), dim3(nblocks), dim3(64), dyn, s, a); else hipLaunchKernelGGL((scan_kernel<false, false, 4, D2, true>), dim3(nblocks), dim3(64), dyn, s, a); }
}

void launch_scan_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, int occ, bool materialise, hipStream_t s) {
    if (nblocks == 0) return;
    size_t dyn = (size_t)(a.lds_pool_elems + a.lds_scr_elems + a.lds_stage_words) * 4 + scan_static_lds();
    if (knob("BVG_SCAN_PAD")) dyn += (size_t)atoi(knob("BVG_SCAN_PAD"));   // occupancy experiments: unused LDS behind the window
    // D2 (lists without reference decoded in place around their intervals): only where such lists exist and are copied from
    const bool d2 = a.min_interval != 0 && a.window > 0 && !(knob("BVG_NO_D2") && atoi(knob("BVG_NO_D2")));
    if (materialise) { if (d2) launch_scan_mat<true>(a, nblocks, wide, dyn, s); else launch_scan_mat<false>(a, nblocks, wide, dyn, s); }
    else if (knob("BVG_SCAN_OCC") && atoi(knob("BVG_SCAN_OCC")) == 5) { if (d2) launch_scan_occ<5, true>(a, nblocks, wide, dyn, s); else launch_scan_occ<5, false>(a, nblocks, wide, dyn, s); }   // experiments: 96 VGPRs, 20 wavefronts per CU
    else if (knob("BVG_SCAN_OCC") && atoi(knob("BVG_SCAN_OCC")) == 50) launch_scan_occ<5, false>(a, nblocks, wide, dyn, s);   // ... without the in-place decode around intervals (14 spills instead of 38)
    else if (occ == 6) launch_scan_occ<6, false>(a, nblocks, wide, dyn, s);
    else if (occ == 5) { if (d2) launch_scan_occ<5, true>(a, nblocks, wide, dyn, s); else launch_scan_occ<5, false>(a, nblocks, wide, dyn, s); }   // 18 / 20 wavefronts per CU
    else if (d2) launch_scan_occ<4, true>(a, nblocks, wide, dyn, s);
    else launch_scan_occ<4, false>(a, nblocks, wide, dyn, s);
}

}  // namespace bvg
