// knn_common.hpp -- pieces shared by the fp32 scan (knn_f32.hip) and the bf16 filter scan (knn_bf16.hip).
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace lemon_knn {

constexpr int BQ = 128;  // query rows per workgroup
constexpr int BX = 128;  // database rows per tile
constexpr int BK = 32;   // k-slice per LDS stage
constexpr int NT = 256;  // threads per workgroup (4 wavefronts, 2x2 over the 128x128 tile)
constexpr int CAP = LEMON_CAND_CAP;

// ---- wave-level selection ---------------------------------------------------------------
// Rank-select: up to 256 distinct keys (4 per lane, 0 = empty) are parked in a per-wave LDS
// scratch; every lane then streams all of them back (broadcast reads) and counts, for each of its
// own keys, how many are larger.  rank < k  <=>  the key is among the k best, and the rank IS its
// position in the sorted output.  No cross-lane dependency chains (the old k-round butterfly
// arg-max was latency bound: ~30k cycles per list; this is ~2.3k VALU instructions).
struct Ranked { int r0, r1, r2, r3; };

__device__ __forceinline__ Ranked wave_rank_keys(u64 v0, u64 v1, u64 v2, u64 v3, int n_bound,
                                                  u64 *__restrict__ sk, int lane) {
    sk[lane] = v0; sk[lane + 64] = v1; sk[lane + 128] = v2; sk[lane + 192] = v3;
    __builtin_amdgcn_wave_barrier();
    Ranked r = {0, 0, 0, 0};
    const int n2 = __builtin_amdgcn_readfirstlane((n_bound + 1) & ~1);
    const ulonglong2 *sk2 = reinterpret_cast<const ulonglong2 *>(sk);
    for (int j = 0; j < n2 / 2; ++j) {
        const ulonglong2 kk = sk2[j];
        r.r0 += (kk.x > v0) + (kk.y > v0);
        r.r1 += (kk.x > v1) + (kk.y > v1);
        r.r2 += (kk.x > v2) + (kk.y > v2);
        r.r3 += (kk.x > v3) + (kk.y > v3);
    }
    __builtin_amdgcn_wave_barrier();
    return r;
}

struct ScanParams {
    const float *qp;      // [nq_pad, dpad] permuted queries, pad rows zero
    const float *xp;      // [n_pad,  dpad] permuted database, pad rows zero
    const float *qnorm;   // [nq_pad]  (L2)
    const float *xnorm;   // [n_pad]   (L2)
    u64 *cand;            // [grid, BQ, CAP]  one region per workgroup
    u64 *part;            // [splits, nq_pad, kk] when splits > 1
    float *D;             // [nq, kk] when splits == 1
    int64_t *I;
    int64_t nq, n;
    int dpad, kk, metric;
    int n_tiles, tiles_per_split, splits;
    int64_t nq_pad;
};

__device__ __forceinline__ int swz(int r, int c) { return r * BK + 4 * (c ^ ((r >> 1) & 7)); }

// global -> register staging of one k-slice: each thread moves 4 16-B chunks per operand.
// (named registers, not arrays: hipcc keeps by-reference float4 arrays in scratch here)
__device__ __forceinline__ float4 stage_ld(const float *__restrict__ src, int dpad, int tid, int i) {
    const int id = tid + NT * i, r = id >> 3, c = id & 7;
    return *reinterpret_cast<const float4 *>(src + (int64_t)r * dpad + 4 * c);
}
__device__ __forceinline__ void stage_st(float *t, int tid, int i, float4 v) {
    const int id = tid + NT * i, r = id >> 3, c = id & 7;
    *reinterpret_cast<float4 *>(&t[swz(r, c)]) = v;
}
#define STAGE_ISSUE(qsrc, xsrc)                                                                     \
    do {                                                                                            \
        rq0 = stage_ld(qsrc, dpad, tid, 0); rq1 = stage_ld(qsrc, dpad, tid, 1);                     \
        rq2 = stage_ld(qsrc, dpad, tid, 2); rq3 = stage_ld(qsrc, dpad, tid, 3);                     \
        rx0 = stage_ld(xsrc, dpad, tid, 0); rx1 = stage_ld(xsrc, dpad, tid, 1);                     \
        rx2 = stage_ld(xsrc, dpad, tid, 2); rx3 = stage_ld(xsrc, dpad, tid, 3);                     \
    } while (0)
#define STAGE_COMMIT(tq_, tx_)                                                                      \
    do {                                                                                            \
        stage_st(tq_, tid, 0, rq0); stage_st(tq_, tid, 1, rq1); stage_st(tq_, tid, 2, rq2);         \
        stage_st(tq_, tid, 3, rq3); stage_st(tx_, tid, 0, rx0); stage_st(tx_, tid, 1, rx1);         \
        stage_st(tx_, tid, 2, rx2); stage_st(tx_, tid, 3, rx3);                                     \
    } while (0)


// ---- lane-private candidate lists ("pair lists") -------------------------------------------------
// In the scans whose accumulator layout puts ONE query on each lane (A operand = database rows,
// B operand = query rows: C[row = db][col = query = lane&31]), a query is owned by the lane pair
// (l, l+32) of one wavefront.  Each of the two lanes appends to its OWN half of the query's list
// (PAIR_CAP/2 entries) with the count in a VGPR: an append is one predicated global store -- no atomic,
// no LDS, no wait.  Thresholds live in VGPRs too.  Entry idx of the pair list: idx < PAIR_CAP/2 -> half 0.
constexpr int PAIR_CAP = 512;

__device__ __forceinline__ u64 pair_load_slot(const u64 *__restrict__ list, int idx, int n0, int n1) {
    const bool ok = idx < PAIR_CAP / 2 ? idx < n0 : (idx - PAIR_CAP / 2) < n1;
    return ok ? list[idx] : 0;
}

// exact selection on exact keys: keep the kk largest keys (by the full 64-bit key: score, then lower
// index), packed at the front of the pair list in arbitrary order: bisection on the key bits (8 compares
// + 8 ballots per lane and step), then ballot-prefix stream compaction.  Needs n0 + n1 >= kk.
// Returns the kk-th largest key.
__device__ __forceinline__ u64 pair_select_exact(u64 *__restrict__ list, int n0, int n1, int kk, int lane) {
    u64 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = pair_load_slot(list, lane + 64 * i, n0, n1);
    // kk-th largest key = largest t with #{key >= t} >= kk.  High word (score) first: 32 steps; the
    // low word (index order among equal scores) only matters when the kk-th score is tied.
    u32 th_hi = 0;
    int c_hi = 0;                                       // #{score-word >= th_hi} at the end (>= kk)
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const u32 cand = th_hi | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) c += __builtin_popcountll(__ballot((u32)(v[i] >> 32) >= cand));
        if (c >= kk) { th_hi = cand; c_hi = c; }        // wave-uniform
    }
    u64 t = (u64)th_hi << 32;                           // all keys with a larger score word are in
    if (th_hi == 0 || c_hi != kk) {                     // tie at the kk-th score (or c_hi never set): refine
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const u64 cand = t | (1ull << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) c += __builtin_popcountll(__ballot(v[i] >= cand));
            if (c >= kk) t = cand;
        }
    } else {
        // exactly kk keys have score word >= th_hi: the kk-th key is the smallest of them
        u64 mn = ~0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i) if ((u32)(v[i] >> 32) >= th_hi && v[i] < mn) mn = v[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(mn, off); mn = o < mn ? o : mn; }
        t = mn;
    }
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int base = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool keep = v[i] >= t && v[i] != 0;
        const u64 m = __ballot(keep);
        if (keep) list[base + __builtin_popcountll(m & below)] = v[i];
        base += __builtin_popcountll(m);
    }
    return t;
}

// sort the best kk keys of a pair list to the front of half-list 0 (best first) by 128-key rank
// merges; returns how many exist (<= kk).  sk: 256-u64 scratch, sb: 64-u64 scratch (per wave, LDS).
__device__ __forceinline__ int pair_sort_topk(u64 *__restrict__ list, int n0, int n1, int kk, int lane,
                                              u64 *__restrict__ sk, u64 *__restrict__ sb) {
    u64 best = 0;
#pragma unroll 1
    for (int base = 0; base < PAIR_CAP; base += 64) {
        if (base < PAIR_CAP / 2 ? base >= n0 : (base - PAIR_CAP / 2) >= n1) continue;   // wave-uniform
        const u64 key = pair_load_slot(list, base + lane, n0, n1);
        const Ranked r = wave_rank_keys(best, key, 0, 0, 128, sk, lane);
        sb[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (best && r.r0 < kk) sb[r.r0] = best;
        if (key && r.r1 < kk) sb[r.r1] = key;
        __builtin_amdgcn_wave_barrier();
        best = sb[lane];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < kk) list[lane] = best;
    return __builtin_popcountll(__ballot(best != 0));
}

// final write-out of a sorted per-row key list (lane < kk holds / reads entry `lane`)
__device__ __forceinline__ void write_out_row(const ScanParams &p, int split, int64_t q, int lane, u64 key) {
    if (lane >= p.kk) return;
    if (p.splits > 1) {
        p.part[((int64_t)split * p.nq_pad + q) * p.kk + lane] = key;
        return;
    }
    float dv; int64_t iv;
    if (key) {
        const float s = lemon_key_score(key);
        dv = (p.metric == LEMON_METRIC_L2) ? -s : s;
        iv = (int64_t)lemon_key_index(key);
    } else {
        dv = (p.metric == LEMON_METRIC_L2) ? FLT_MAX : -FLT_MAX;
        iv = -1;
    }
    p.D[q * p.kk + lane] = dv;
    p.I[q * p.kk + lane] = iv;
}

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace lemon_knn

// host helpers implemented in knn_f32.hip
int lemon_permute_rows(const float *src, int64_t n, int d, float *dst, int dpad, hipStream_t s);
int lemon_launch_merge(const u64 *part, int splits, int64_t nq_pad, int64_t nq, int kk, int metric, float *D,
                       int64_t *I, hipStream_t stream);
int lemon_fill_empty(float *D, int64_t *I, int64_t total, int metric, hipStream_t stream);
void lemon_plan_splits(int panels, int n_tiles, int *splits, int *tiles_per_split);
int lemon_ensure_search_ws(lemon_index_t *idx, int64_t nq_pad, int splits, int qp_row_bytes, int cand_cap,
                           hipStream_t stream);
