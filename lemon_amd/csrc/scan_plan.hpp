// scan_plan.hpp -- the host-side segment planner of the exact scan (knn_f32.hip): plain C++, no HIP, so that it can also be
// compiled with gcc under AddressSanitizer / UBSan and fuzzed on the CPU (tests/native/plan_fuzz.cpp, tests/test_sanitizers.py).
#pragma once
#include <stdint.h>
#include <algorithm>
#include <utility>
#include <vector>

struct LemonPlan {
    std::vector<int> seg_begin;      // [grid + 1]
    std::vector<int> pieces;         // [panels]
    std::vector<int> segs;           // 4 ints per segment
    int grid = 0, splits = 1;
};

// `c` = what a segment costs on top of its tiles (pipeline refill, the cold start of its candidate lists, the final
// selection), in tile times: shares are equal in COST, not in tiles -- a workgroup that collects four tails would
// otherwise finish 5 % after one that walks a single head (measured: lives 16.9 ms against 16.0-16.3 ms).
static inline void plan_group(const std::vector<int> &wgs, const std::vector<int> &pnls, int T, int c, std::vector<std::vector<int>> &wg_segs,
                       std::vector<int> &pieces) {
    const int m = (int)wgs.size(), n = (int)pnls.size();
    if (m == 0 || n == 0) return;
    const int64_t total = (int64_t)n * T;
    const int64_t u = (total + m - 1) / m;
    // budget per workgroup: tiles + c per segment; about one segment per panel and one more per workgroup boundary
    int64_t budget;
    std::vector<int64_t> room;
    auto put = [&](int w, int panel, int t0, int nt) {
        std::vector<int> &v = wg_segs[wgs[w]];
        v.push_back(panel); v.push_back(t0); v.push_back(nt); v.push_back(pieces[panel]++);
        room[w] -= nt + c;
    };
    std::vector<std::pair<int, int>> pool;   // (panel, first tile) of what the aligned part leaves, panel-major
    if ((int64_t)T >= u) {                   // heads [0, h) of one panel per workgroup, tails to the pool
        int64_t h = (total + (int64_t)c * n + m - 1) / m;      // h + c = (n (T - h) + c m) / (m - n)
        if (h > T || n == m) h = T;
        budget = h + c;
        room.assign(m, budget);
        for (int i = 0; i < n; ++i) {
            put(i, pnls[i], 0, (int)h);      // n <= m here (n T <= m u and T >= u)
            if ((int64_t)T > h) pool.push_back(std::make_pair(pnls[i], (int)h));
        }
    } else {                                 // whole panels per workgroup while they fit the budget, the rest to the pool
        budget = ((int64_t)n * (T + c) + m - 1) / m + c;
        room.assign(m, budget);
        const int a = (int)(budget / (T + c)) > 0 ? (int)(budget / (T + c)) : 1;
        int next = 0;
        for (int w = 0; w < m; ++w)
            for (int j = 0; j < a && next < n; ++j) put(w, pnls[next++], 0, T);
        for (; next < n; ++next) pool.push_back(std::make_pair(pnls[next], 0));
    }
    int w = 0;
    for (auto &pt : pool) {                  // share the pool out in order: every workgroup is filled up to its budget
        int t0 = pt.second;
        while (t0 < T) {
            while (w < m - 1 && room[w] <= c) ++w;
            int64_t fit = room[w] - c;
            if (fit < 1 || w == m - 1) fit = T;              // the last workgroup takes whatever is left
            const int nt = (int)std::min<int64_t>(fit, T - t0);
            put(w, pt.first, t0, nt);
            t0 += nt;
        }
    }
}

// the plan for `slots` resident workgroup slots (2 per CU), `xcds` XCD groups (1 = not XCD-aware) and `seg_cost` tile times per segment
static inline void lemon_plan_segments_host(int panels, int n_tiles, int slots, int xcds, int seg_cost, LemonPlan &plan) {
    const int64_t units = (int64_t)panels * n_tiles;
    int64_t g = units / 8;                               // every workgroup keeps >= 8 tiles
    if (g < 1) g = 1;
    if (g > slots) g = slots;
    const int X = (panels >= 8 * xcds && g >= 8 * xcds) ? xcds : 1;   // few panels: one group, still aligned
    std::vector<std::vector<int>> wg_segs((size_t)g);
    plan.pieces.assign((size_t)panels, 0);
    for (int x = 0; x < X; ++x) {
        std::vector<int> wgs, pnls;
        for (int b = x; b < (int)g; b += X) wgs.push_back(b);
        for (int q = x; q < panels; q += X) pnls.push_back(q);
        plan_group(wgs, pnls, n_tiles, seg_cost, wg_segs, plan.pieces);
    }
    plan.grid = (int)g;
    plan.seg_begin.assign((size_t)g + 1, 0);
    plan.segs.clear();
    for (int b = 0; b < (int)g; ++b) {
        plan.seg_begin[b] = (int)(plan.segs.size() / 4);
        plan.segs.insert(plan.segs.end(), wg_segs[b].begin(), wg_segs[b].end());
    }
    plan.seg_begin[g] = (int)(plan.segs.size() / 4);
    plan.splits = 1;
    for (int c : plan.pieces) plan.splits = c > plan.splits ? c : plan.splits;
}
