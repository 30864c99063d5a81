// C++ twin of BVGraphTest.testLarge / WebGraphTestCase.assertGraph over the host mirror
// (webgraph-big_amd/host/bvgraph.hpp -> C ABI -> HIP kernels).  Prints what it measured; the pytest
// wrapper (tests/test_gpu_cpp_mirror.py) compares with the oracle / golden.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../webgraph-big_amd/host/bvgraph.hpp"

using namespace webgraph;

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s basename\n", argv[0]); return 2; }
    try {
        auto g = BVGraph::load(argv[1]);
        const int64_t n = g->numNodes();
        uint64_t arcs = 0, chk = 0;
        auto it = g->nodeIterator();
        bool threw = false;
        try { it.outdegree(); } catch (const std::logic_error&) { threw = true; }          // IllegalStateException before nextLong
        if (!threw) { printf("FAIL no IllegalStateException\n"); return 1; }
        while (it.hasNext()) {
            int64_t x = it.nextLong();
            int64_t d = it.outdegree();
            auto s = it.successors();
            int64_t prev = -1;
            for (int64_t j = 0; j < d; j++) {
                int64_t y = s.nextLong();
                if (y <= prev) { printf("FAIL not increasing at node %lld\n", (long long)x); return 1; }
                prev = y;
                chk += bvg_arc_mix((uint64_t)x, (uint64_t)y);
            }
            if (s.nextLong() != -1) { printf("FAIL no -1 after %lld successors\n", (long long)d); return 1; }
            arcs += (uint64_t)d;
        }
        threw = false;
        try { it.nextLong(); } catch (const std::out_of_range&) { threw = true; }            // NoSuchElementException
        if (!threw) { printf("FAIL no NoSuchElementException\n"); return 1; }
        // random access agrees with the sequential scan on a few nodes; copies are independent
        auto g2 = g->copy();
        auto it2 = g2->nodeIterator(n / 2);
        for (int i = 0; i < 50 && it2.hasNext(); i++) {
            int64_t x = it2.nextLong();
            auto ra = g->successorBigArray(x);
            if ((int64_t)ra.size() != it2.outdegree() || g->outdegree(x) != it2.outdegree()) { printf("FAIL outdegree mismatch\n"); return 1; }
            const int64_t* sa = it2.successorBigArray();
            for (size_t j = 0; j < ra.size(); j++) if (ra[j] != sa[j]) { printf("FAIL successor mismatch\n"); return 1; }
        }
        {   // frontier-style batched random access
            std::vector<int64_t> fr = {0, n - 1, n / 2, 5, 5, n / 3};
            std::vector<int32_t> fd; std::vector<int64_t> fs;
            g->successorsBatch(fr, fd, fs);
            size_t o = 0;
            for (size_t q = 0; q < fr.size(); q++) {
                auto ra = g->successorBigArray(fr[q]);
                if ((size_t)fd[q] != ra.size()) { printf("FAIL batch outdegree\n"); return 1; }
                for (size_t j = 0; j < ra.size(); j++) if (fs[o + j] != ra[j]) { printf("FAIL batch successors\n"); return 1; }
                o += ra.size();
            }
        }
        {   // transposition feed: every arc (x,y) must appear as x in y's incoming list, lists increasing, same arc count
            std::vector<uint64_t> toff; std::vector<int64_t> ts;
            g->transposeCSR(toff, ts);
            if (toff.size() != (size_t)n + 1 || toff[(size_t)n] != arcs || ts.size() != arcs) { printf("FAIL transpose size\n"); return 1; }
            for (int64_t y = 0; y < n; y++) for (uint64_t t = toff[(size_t)y] + 1; t < toff[(size_t)y + 1]; t++) if (ts[t - 1] >= ts[t]) { printf("FAIL transpose order\n"); return 1; }
            for (int64_t x : {(int64_t)0, n / 2, n - 1}) {
                for (int64_t y : g->successorBigArray(x)) {
                    bool found = false;
                    for (uint64_t t = toff[(size_t)y]; t < toff[(size_t)y + 1] && !found; t++) found = ts[t] == x;
                    if (!found) { printf("FAIL transpose arc\n"); return 1; }
                }
            }
        }
        {   // symmetrisation: sizes, order, and symmetry of a few nodes' lists
            std::vector<uint64_t> so; std::vector<int64_t> ss;
            g->symmetrizeCSR(so, ss);
            if (so.size() != (size_t)n + 1 || so[(size_t)n] != ss.size() || ss.size() < arcs || ss.size() > 2 * arcs) { printf("FAIL symmetrize size\n"); return 1; }
            for (int64_t x : {(int64_t)0, n / 2, n - 1}) for (uint64_t t = so[(size_t)x]; t < so[(size_t)x + 1]; t++) {
                if (t > so[(size_t)x] && ss[t - 1] >= ss[t]) { printf("FAIL symmetrize order\n"); return 1; }
                const int64_t y = ss[t]; bool found = false;
                for (uint64_t u = so[(size_t)y]; u < so[(size_t)y + 1] && !found; u++) found = ss[u] == x;
                if (!found) { printf("FAIL symmetrize symmetry\n"); return 1; }
            }
        }
        threw = false;
        try { g->outdegree(n); } catch (const std::invalid_argument&) { threw = true; }
        if (!threw) { printf("FAIL no IllegalArgumentException\n"); return 1; }
        uint64_t parts = 0;
        for (auto& p : g->splitNodeIterators(4)) while (p.hasNext()) { p.nextLong(); parts += (uint64_t)p.outdegree(); }
        bvg_scan_result r = g->scan();
        if (argc > 2) {   // labelled twin: labels in successor order, every label = (source * 31 + position in the list) & 1023
            auto lg = BitStreamArcLabelledImmutableGraph::load(argv[2]);
            std::vector<int32_t> d, lab; std::vector<int64_t> sc;
            lg->decodeRange(0, lg->numNodes(), d, sc, lab);
            size_t o = 0;
            for (int64_t x = 0; x < lg->numNodes(); x++) for (int32_t j = 0; j < d[(size_t)x]; j++, o++)
                if (lab[o] != (int32_t)((x * 31 + j) & 1023)) { printf("FAIL label of arc %lld/%d\n", (long long)x, j); return 1; }
            if (o != lab.size()) { printf("FAIL label count\n"); return 1; }
            printf("LABELS %zu ok\n", o);
        }
        if (n <= 2000000) {   // BVGraph.store on the device gives back the very file that was loaded (the reference's own fixture in the pytest run)
            std::vector<int32_t> d; std::vector<int64_t> sc;
            g->decodeRange(0, n, d, sc);
            std::vector<uint64_t> off((size_t)n + 1, 0);
            for (int64_t x = 0; x < n; x++) off[(size_t)x + 1] = off[(size_t)x] + (uint64_t)d[(size_t)x];
            std::vector<uint8_t> bytes; std::vector<uint64_t> offs;
            BVGraph::store(g->params(), off, sc, bytes, offs);
            FILE* f = fopen((std::string(argv[1]) + ".graph").c_str(), "rb");
            if (!f) { printf("FAIL cannot reopen the .graph file\n"); return 1; }
            std::vector<uint8_t> file; int c; while ((c = fgetc(f)) != EOF) file.push_back((uint8_t)c);
            fclose(f);
            if (file != bytes) { printf("FAIL store: %zu bytes, the file has %zu\n", bytes.size(), file.size()); return 1; }
            printf("STORE %zu bytes identical\n", bytes.size());
        }
        printf("OK nodes=%lld arcs=%llu chk=%016llx scan_arcs=%llu scan_chk=%016llx split_arcs=%llu\n", (long long)n, (unsigned long long)arcs,
               (unsigned long long)chk, (unsigned long long)r.arcs, (unsigned long long)r.chk, (unsigned long long)parts);
        return 0;
    } catch (const std::exception& e) {
        printf("FAIL exception: %s\n", e.what());
        return 1;
    }
}
