// plan_fuzz.cpp -- property fuzz of the exact scan's segment planner (lemon_amd/csrc/scan_plan.hpp) on the CPU, built with
// gcc under AddressSanitizer + UBSan by tests/test_sanitizers.py (GPU sanitizers are not available on the pool; the planner is
// host code).  For random (panels, tiles, slots, xcds, seg_cost): every (panel, tile) unit is covered exactly once, no segment
// leaves its panel or the grid, a panel's pieces are numbered 0 .. pieces-1 without gaps, seg_begin is monotone.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tests/native/plan_fuzz.cpp -o plan_fuzz && ./plan_fuzz [cases] [seed]
#include "../../lemon_amd/csrc/scan_plan.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>

static int check(int panels, int tiles, int slots, int xcds, int cost) {
    LemonPlan plan;
    lemon_plan_segments_host(panels, tiles, slots, xcds, cost, plan);
    const int grid = plan.grid, nseg = (int)(plan.segs.size() / 4);
#define FAIL(msg) do { fprintf(stderr, "FAIL panels=%d tiles=%d slots=%d xcds=%d cost=%d: %s\n", panels, tiles, slots, xcds, cost, msg); return 1; } while (0)
    if (grid < 1 || grid > slots) FAIL("grid outside [1, slots]");
    if ((int)plan.seg_begin.size() != grid + 1 || plan.seg_begin[0] != 0 || plan.seg_begin[grid] != nseg) FAIL("seg_begin ends");
    for (int b = 0; b < grid; ++b) if (plan.seg_begin[b + 1] < plan.seg_begin[b]) FAIL("seg_begin not monotone");
    if ((int)plan.pieces.size() != panels) FAIL("pieces size");
    std::vector<unsigned char> cover((size_t)panels * tiles, 0);
    std::vector<std::vector<unsigned char>> seen(panels);
    for (int p = 0; p < panels; ++p) seen[p].assign((size_t)std::max(plan.pieces[p], 0), 0);
    int splits = 1;
    for (int i = 0; i < nseg; ++i) {
        const int p = plan.segs[4 * i], t0 = plan.segs[4 * i + 1], nt = plan.segs[4 * i + 2], piece = plan.segs[4 * i + 3];
        if (p < 0 || p >= panels || t0 < 0 || nt < 1 || (int64_t)t0 + nt > tiles) FAIL("segment outside its panel");
        if (piece < 0 || piece >= plan.pieces[p] || seen[p][piece]) FAIL("piece number");
        seen[p][piece] = 1;
        for (int t = t0; t < t0 + nt; ++t) if (cover[(size_t)p * tiles + t]++) FAIL("unit covered twice");
    }
    for (size_t u = 0; u < cover.size(); ++u) if (cover[u] != 1) FAIL("unit not covered");
    for (int p = 0; p < panels; ++p) {
        for (unsigned char s : seen[p]) if (!s) FAIL("gap in a panel's piece numbers");
        splits = std::max(splits, plan.pieces[p]);
    }
    if (splits != plan.splits) FAIL("splits != max pieces");
    return 0;
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 12345);
    auto pick = [&](int lo, int hi) { return (int)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    int bad = 0;
    // corners first
    const int corner[][5] = {{1, 1, 512, 8, 3}, {1, 1, 1, 1, 0}, {391, 313, 512, 8, 3}, {7813, 7813, 512, 8, 3}, {64, 8, 512, 8, 3}, {63, 9, 7, 8, 3},
                             {4096, 1, 512, 8, 3}, {1, 100000, 512, 8, 3}, {65, 65, 64, 8, 0}, {512, 64, 512, 1, 3}, {3, 5, 2, 2, 50}};
    for (auto &c : corner) bad += check(c[0], c[1], c[2], c[3], c[4]);
    for (int i = 0; i < cases && !bad; ++i) {
        const int kind = pick(0, 3);
        const int panels = kind == 0 ? pick(1, 40) : kind == 1 ? pick(1, 600) : pick(1, 3000);
        const int tiles = kind == 0 ? pick(1, 60) : kind == 2 ? pick(1, 200) : pick(1, 2500);
        bad += check(panels, tiles, pick(1, 3) == 1 ? pick(1, 40) : 64 * pick(1, 16), pick(0, 2) ? 8 : pick(1, 8), pick(0, 6));
    }
    printf("plan_fuzz: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
