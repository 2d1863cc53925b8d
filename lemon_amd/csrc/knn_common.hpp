// knn_common.hpp -- pieces shared by the fp32 scan (knn_f32.hip) and the bf16 filter scan (knn_bf16.hip).
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace lemon_knn {

// three-input maximum in ONE instruction (fmaxf() makes hipcc quiet every input with `v_max x, x` first; matrix-core
// results are never signalling NaNs, and a quiet NaN operand is ignored here exactly as a '>' test ignores it)
__device__ __forceinline__ float max3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

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
    unsigned long long *phase_dbg;   // diagnostic instantiation only: 4 cycle sums
    int ablate;           // diagnostic instantiation only: 1 = no operand loads, 2 = no barriers (results invalid)
    int stale;            // new entries per query that trigger a re-selection (fp32 scan)
    int fair;             // fp32 scan: the co-resident workgroups of a CU take turns at the higher issue priority
    const int *plan;      // planned decomposition (knn_f32.hip, lemon_plan_segments): [grid + 1] first segment of every
                          // workgroup, ..., then at int offset plan_segs the segments (panel, first tile, tiles, piece);
                          // nullptr: one (panel, split) rectangle per workgroup.  splits = max pieces per panel
    int plan_segs;
    unsigned *th_pub;     // [nq_pad] order-encoded admission bounds the workgroups sharing a query panel publish to each
                          // other (zero-initialised per launch), or nullptr: see k_scan_f32 "shared bounds"
    const u64 *ub;        // [nq_pad] exclusive upper bound on the key of an admissible row, or nullptr (k > 64: the
                          // passes after the first only admit rows ranked behind the previous pass's last result)
};

__device__ __forceinline__ int swz(int r, int c) { return r * BK + 4 * (c ^ ((r >> 1) & 7)); }

// global -> register staging of one k-slice: each thread moves 4 16-B chunks per operand.
// (named registers, not arrays: hipcc keeps by-reference float4 arrays in scratch here)
// `src` is wave-uniform, `voff` = ((tid>>3)*dpad + 4*(tid&7))*4 the thread's byte offset inside a 32-row
// block: the load becomes SGPR base + 32-bit VGPR offset (no per-lane 64-bit address arithmetic).
__device__ __forceinline__ float4 stage_ld(const float *__restrict__ src, int dpad, unsigned voff, int i) {
    const char *b = reinterpret_cast<const char *>(src + (int64_t)(NT / 8) * i * dpad);
    return *reinterpret_cast<const float4 *>(b + voff);
}
__device__ __forceinline__ void stage_st(float *t, int tid, int i, float4 v) {
    const int id = tid + NT * i, r = id >> 3, c = id & 7;
    *reinterpret_cast<float4 *>(&t[swz(r, c)]) = v;
}
#define STAGE_ISSUE(qsrc, xsrc)                                                                     \
    do {                                                                                            \
        rq0 = stage_ld(qsrc, dpad, voff, 0); rq1 = stage_ld(qsrc, dpad, voff, 1);                   \
        rq2 = stage_ld(qsrc, dpad, voff, 2); rq3 = stage_ld(qsrc, dpad, voff, 3);                   \
        rx0 = stage_ld(xsrc, dpad, voff, 0); rx1 = stage_ld(xsrc, dpad, voff, 1);                   \
        rx2 = stage_ld(xsrc, dpad, voff, 2); rx3 = stage_ld(xsrc, dpad, voff, 3);                   \
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

// Exact selection on exact keys.  The pair list is read into NS register slots (64 keys each; the
// occupied 64-blocks of half 0 first, then those of half 1 -- typical lists fill 2-3 of the 8 possible
// slots, and the bisection below costs one compare + ballot per slot and step, so NS is specialised).
// kk-th largest key = largest t with #{key >= t} >= kk, found by bisection on the key bits: the high
// word (score) first, 32 steps; the low word (index order among equal scores) only when the kk-th
// score is tied.  Fewer than kk keys: t = 0 (everything survives).
template <int NS>
struct PairSlots {
    u64 v[NS];
    __device__ __forceinline__ void load(const u64 *__restrict__ list, int n0, int n1, int s0, int lane) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const bool lo = j < s0;
            const int pos = lane + 64 * (lo ? j : j - s0);
            const bool ok = lo ? pos < n0 : pos < n1;
            v[j] = ok ? list[lo ? pos : PAIR_CAP / 2 + pos] : 0;
        }
    }
    __device__ __forceinline__ u64 kth(int kk) const {
        u32 th_hi = 0;
        int c_hi = 0;                                   // #{score-word >= th_hi} at the end (>= kk)
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const u32 cand = th_hi | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < NS; ++i) c += __builtin_popcountll(__ballot((u32)(v[i] >> 32) >= cand));
            if (c >= kk) {                              // wave-uniform
                th_hi = cand; c_hi = c;
                if (c == kk) break;                     // exactly kk keys reach this prefix: they ARE the top kk
            }
        }
        u64 t = (u64)th_hi << 32;                       // all keys with a larger score word are in
        if (th_hi == 0 || c_hi != kk) {                 // tie at the kk-th score (or c_hi never set): refine
#pragma unroll 1
            for (int bit = 31; bit >= 0; --bit) {
                const u64 cand = t | (1ull << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < NS; ++i) c += __builtin_popcountll(__ballot(v[i] >= cand));
                if (c >= kk) t = cand;
            }
        } else {
            // exactly kk keys have score word >= th_hi: the kk-th key is the smallest of them
            u64 mn = ~0ull;
#pragma unroll
            for (int i = 0; i < NS; ++i) if ((u32)(v[i] >> 32) >= th_hi && v[i] < mn) mn = v[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(mn, off); mn = o < mn ? o : mn; }
            t = mn;
        }
        return t;
    }
    // Maintenance only needs a LOWER BOUND of the kk-th key that keeps the list short: bisect the score word from the
    // top and stop at the first prefix with kk <= #{score word >= prefix} <= kk + slack (typically ~14 of the 32 steps:
    // sign + exponent are shared, a few mantissa bits separate ~150 candidates).  Everything >= the prefix survives.
    // Only when the score word cannot separate them (exact duplicates at the kk-th score) the exact 64-bit key is found.
    __device__ __forceinline__ u64 kth_loose(int kk, int slack) const {
        u32 th_hi = 0;
        int c_hi = 0x7fffffff;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const u32 cand = th_hi | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < NS; ++i) c += __builtin_popcountll(__ballot((u32)(v[i] >> 32) >= cand));
            if (c >= kk) {                              // wave-uniform
                th_hi = cand; c_hi = c;
                if (c <= kk + slack) break;
            }
        }
        u64 t = (u64)th_hi << 32;
        if (c_hi > kk + slack) {                        // tie at the kk-th score word: refine on the index word
#pragma unroll 1
            for (int bit = 31; bit >= 0; --bit) {
                const u64 cand = t | (1ull << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < NS; ++i) c += __builtin_popcountll(__ballot(v[i] >= cand));
                if (c >= kk) t = cand;
            }
        }
        return t;
    }
    // survivors (key >= t, non-empty) packed to dst[0..) in arbitrary order; returns their count
    __device__ __forceinline__ int compact(u64 *__restrict__ dst, u64 t, int lane) const {
        const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        int base = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const bool keep = v[i] >= t && v[i] != 0;
            const u64 m = __ballot(keep);
            if (keep) dst[base + __builtin_popcountll(m & below)] = v[i];
            base += __builtin_popcountll(m);
        }
        return base;
    }
};

// maintenance: keep the best keys -- at least the kk largest, at most kk + SELECT_SLACK unless the kk-th score is
// tied -- packed at the front of half-list 0; returns a lower bound t of the kk-th largest key (every key >= t was
// kept) and the number kept in *kept.  Needs n0 + n1 >= kk.
constexpr int SELECT_SLACK = 24;
template <int NS>
__device__ __forceinline__ u64 pair_select_ns(u64 *__restrict__ list, int n0, int n1, int s0, int kk, int lane, int *kept) {
    PairSlots<NS> ps;
    ps.load(list, n0, n1, s0, lane);
    const u64 t = ps.kth_loose(kk, SELECT_SLACK);
    *kept = ps.compact(list, t, lane);                  // all loads above precede these stores
    return t;
}
__device__ __forceinline__ u64 pair_select_loose(u64 *__restrict__ list, int n0, int n1, int kk, int lane, int *kept) {
    const int s0 = (n0 + 63) >> 6, ns = s0 + ((n1 + 63) >> 6);   // wave-uniform
    if (ns <= 2) return pair_select_ns<2>(list, n0, n1, s0, kk, lane, kept);
    if (ns == 3) return pair_select_ns<3>(list, n0, n1, s0, kk, lane, kept);
    if (ns == 4) return pair_select_ns<4>(list, n0, n1, s0, kk, lane, kept);
    return pair_select_ns<8>(list, n0, n1, s0, kk, lane, kept);
}

// final: the best kk keys of a pair list, sorted: lane's return value is one surviving key (0 = none)
// and *pos its position in best-first order (keys are distinct, so ranks are a permutation); lanes
// without a key get *pos = lane.  sk: >= 64-u64 per-wave LDS scratch.
// SORT = false (a partial list that k_merge rank-selects anyway): the best kk in arbitrary order, *pos = lane.
template <int NS, bool SORT>
__device__ __forceinline__ u64 pair_final_ns(const u64 *__restrict__ list, int n0, int n1, int s0, int kk,
                                             int lane, u64 *__restrict__ sk, int *pos) {
    PairSlots<NS> ps;
    ps.load(list, n0, n1, s0, lane);
    const u64 t = ps.kth(kk);
    __builtin_amdgcn_wave_barrier();
    const int cnt = ps.compact(sk, t, lane);            // <= kk <= 64 survivors
    __builtin_amdgcn_wave_barrier();
    const u64 mine = lane < cnt ? sk[lane] : 0;
    __builtin_amdgcn_wave_barrier();
    if (!SORT) { *pos = lane; return mine; }
    int r = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const u32 lo = __builtin_amdgcn_readlane((u32)mine, j), hi = __builtin_amdgcn_readlane((u32)(mine >> 32), j);
        r += (((u64)hi << 32) | lo) > mine;
    }
    *pos = mine ? r : lane;
    return mine;
}
template <bool SORT>
__device__ __forceinline__ u64 pair_final_topk(const u64 *__restrict__ list, int n0, int n1, int kk, int lane,
                                               u64 *__restrict__ sk, int *pos) {
    const int s0 = (n0 + 63) >> 6, ns = s0 + ((n1 + 63) >> 6);
    if (ns <= 2) return pair_final_ns<2, SORT>(list, n0, n1, s0, kk, lane, sk, pos);
    if (ns <= 4) return pair_final_ns<4, SORT>(list, n0, n1, s0, kk, lane, sk, pos);
    return pair_final_ns<8, SORT>(list, n0, n1, s0, kk, lane, sk, pos);
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

// slot of the next append inside a lane's half-list.  The scans keep cnt <= PAIR_CAP/2 - BX/2 before a tile (the `full`
// trigger) and a tile adds at most BX/2 entries per lane, so cnt < PAIR_CAP/2 holds at every append; the clamp makes a
// broken invariant (a diagnostic build that skips maintenance did that once: DESIGN.md "The recorded GPU fault") overwrite
// the half-list's last slot instead of leaving the candidate workspace.
__device__ __forceinline__ unsigned append_slot(int cnt) {
    return (unsigned)cnt < (unsigned)(PAIR_CAP / 2 - 1) ? (unsigned)cnt : (unsigned)(PAIR_CAP / 2 - 1);
}

}  // namespace lemon_knn

// host helpers implemented in knn_f32.hip
int lemon_permute_rows(const float *src, int64_t n, int d, float *dst, int dpad, hipStream_t s);
int lemon_launch_merge(const u64 *part, int splits, const int *pieces, int64_t nq_pad, int64_t nq, int kk, int metric, float *D,
                       int64_t *I, hipStream_t stream);
int lemon_fill_empty(float *D, int64_t *I, int64_t total, int metric, hipStream_t stream);
void lemon_plan_splits(int panels, int n_tiles, int *splits, int *tiles_per_split);
int lemon_ensure_search_ws(lemon_index_t *idx, int64_t nq_pad, int splits, int64_t n_wg, int qp_row_bytes, int cand_cap,
                           hipStream_t stream);
// LEMON_ABLATE (diagnostics): the value must be a combination of the bits `allowed` names for this kernel, anything else
// is refused (LEMON_E_INVALID) instead of silently selecting whatever code a stray bit happens to reach
int lemon_parse_ablate(const char *kernel, int allowed, int *out);
