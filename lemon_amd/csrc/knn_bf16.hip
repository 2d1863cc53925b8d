// knn_bf16.hip -- LEMON_ALGO_BF16_FILTER: 16-bit MFMA filter scan + exact float32 re-rank.
//
// The tile product runs on the 16-bit matrix pipe (v_mfma_f32_32x32x16_f16: 16x the fp32 MFMA rate, half the staged bytes)
// over 16-bit (round-to-nearest-even) copies of Q and X -- fp16 since round 3, bf16 before; file, algorithm and kernel names
// kept.  The approximate score s~ is only a FILTER with a proven error bound
//     |s~(q,x) - s(q,x)| <= eps(q) := ||q||*max_j||x_j - xh_j|| + ||q - qh||*max_j||xh_j||     (rounding)
//                                    + 3*d*2^-24 * ||q||*max_j||x_j||                          (fp32 sums)
// (identity  sum q x - sum qh xh = sum q (x-xh) + sum (q-qh) xh  + Cauchy-Schwarz; the residual norms
// are MEASURED when the 16-bit copies are made: 1.4e-4 of the norms for fp16 on embedding data, 1.1e-3 for bf16),
// where s is the float32 fmaf-chain score of the numeric contract.
//
// Candidate bookkeeping is the fp32 scan's (keys = ord(score)<<32 | ~index, per-query lists, rank-select
// compaction) but on APPROXIMATE keys: with tau = the k-th largest s~ seen so far, k rows have exact
// score >= tau - eps, so a row can only be in the top-k if s~ + eps >= tau - eps.  Rows with
// s~ <= tau - 2 eps are dropped (at the tile epilogue, or at a "light" compaction, which needs no
// database access and keeps the thresholds fresh); everything else survives to ONE exact re-scoring
// pass at the end (one lane per row walks the fp32 chain over the original data), followed by an exact
// rank-select.  If the band holds more rows than a list can carry (concentrated data), the list is
// re-scored exactly on the spot and cut to k.  The band is a proof, so there is no fallback path and the
// output is bit-identical to LEMON_ALGO_F32_MFMA and the CPU oracle.
#include "knn_common.hpp"
#include <stdlib.h>

using namespace lemon_knn;

// The 16-bit filter format is IEEE fp16 (11 significant bits; the first two rounds used bf16, whose 8 bits made the band 5x
// wider -- the names of the algorithm and of the kernels still say bf16).  Unit-norm embeddings sit far inside its range; the
// copy kernel saturates at +-65 504 and flushes fp16 subnormals to zero, and whatever that costs in accuracy is in the MEASURED
// residual the band is built from, so the proof does not depend on the format.
typedef _Float16 lp16;
typedef lp16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BKH = 64;   // bf16 k-slice per LDS stage (128 B rows, like fp32 BK=32)
constexpr int CAPH = 512; // candidate slots per query (approximate keys): 8 per lane in a light compaction
static_assert(CAPH == PAIR_CAP, "append_slot() clamps to the pair-list half capacity");
constexpr int REFRESH = 96; // light-compact a list after this many new candidates: the admission bound then
                            // tracks the running k-th best closely (appends ~ k ln(N/k) instead of 3-4x that)

// f32 [n,d] -> bf16 [*, dpad_h] (RNE, zero padded columns), one wavefront per row, plus the row's
// measured rounding residual ||x - xh||^2 and ||xh||^2 (float32 sums; consumers inflate them)
__global__ __launch_bounds__(256) void k_convert_bf16(const float *__restrict__ src, int64_t n, int d,
                                                      lp16 *__restrict__ dst, int dpad_h,
                                                      float *__restrict__ res2, float *__restrict__ hn2) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float *s = src + r * (int64_t)d;
    float e2 = 0.0f, h2 = 0.0f;
    for (int u = lane; u < dpad_h / 8; u += 64) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (8 * u + e < d) ? s[8 * u + e] : 0.0f;
            lp16 b = (lp16)fminf(fmaxf(v, -65504.0f), 65504.0f);      // saturate: the residual below then carries the excess
            float vb = (float)b;
            if (fabsf(vb) < 6.103515625e-05f) { b = (lp16)0.0f; vb = 0.0f; }   // no fp16 subnormals into the matrix pipe
            const float dv = v - vb;
            e2 = __builtin_fmaf(dv, dv, e2);
            h2 = __builtin_fmaf(vb, vb, h2);
            o[e] = b;
        }
        *reinterpret_cast<bf16x8 *>(dst + r * (int64_t)dpad_h + 8 * u) = o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e2 += __shfl_xor(e2, off); h2 += __shfl_xor(h2, off); }
    if (lane == 0) { res2[r] = e2; hn2[r] = h2; }
}

// max of non-negative floats through their bit patterns (NaN -> +inf: the band then admits everything)
__global__ __launch_bounds__(256) void k_max_nonneg(const float *__restrict__ v, int64_t n, unsigned *__restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = v[i];
        m = (x > m || x != x) ? x : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m != m ? INFINITY : m));
}

struct ScanParamsH {
    ScanParams b;              // qp / xp unused here
    const lp16 *qh, *xh;     // [nq_pad, dpad_h], [n_pad, dpad_h]
    const float *q, *x;        // originals, row-major [nq, d], [n, d]
    const float *qres2, *qhn2; // [nq_pad] measured ||q-qh||^2, ||qh||^2
    const unsigned *xstat;     // device scalars (float bits): [0] max dot(x,x), [1] max ||x-xh||^2, [2] max ||xh||^2
    int d, dpad_h;
    // database chunking (Q-stationary kernel, splits == 1): one launch scans tiles [chunk_t0, chunk_t1)
    // for every query panel, so that the chunk is re-read from the 256 MiB Infinity Cache instead of
    // HBM; per-query state (list length, admission bound) lives in `state` between launches
    int chunk_t0, chunk_t1, first_chunk, last_chunk;
    int *cnt;                  // [grid * queries per workgroup][2]: half-list counts at the end of the scan, for k_bf16_final
    float *state;              // [grid*256 lanes][4]: {half-list count, last compacted length, thr_key, -}
    int ablate;                // diagnostics only (LEMON_ABLATE): 1 = skip the filter epilogue AND maintenance (nothing is
                               // appended), 4 = nothing passes the filter, 8 (phase-stamped build) = the tile-end wait for
                               // the DMA is booked under "maintain", so that "sync" is the barrier alone (measured: 2.6 %
                               // and 8.7 % of wave 0's cycles).  There is deliberately NO "appends without maintenance"
                               // mode: maintenance is what keeps the 256-entry half-lists in bounds (a working-tree
                               // diagnostic of round 2 had one as value 2 and faulted the GPU; lemon_parse_ablate now
                               // refuses unknown bits and append_slot() clamps the store)
    unsigned long long *phase_dbg;   // diagnostic builds only: [grid][4] cycle sums (loop, epilogue, sync, maintain)
};

// proven bound on |s~ - s| for one query (see file header); for L2 the bound on the key -D
__device__ __forceinline__ float band_eps_raw(const unsigned *__restrict__ xstat, int d, float qn, float qres2, bool l2) {
    const float xn2 = __uint_as_float(xstat[0]), xr2 = __uint_as_float(xstat[1]), xh2 = __uint_as_float(xstat[2]);
    const float nq = sqrtf(qn) * 1.0005f;
    float eps = (nq * sqrtf(xr2) + sqrtf(qres2) * sqrtf(xh2)) * 1.002f
              + 3.0f * (float)d * 5.9604645e-8f * nq * sqrtf(xn2) * 1.002f + 1e-30f;
    if (l2) eps = 2.0f * eps + 4.8e-7f * (qn + xn2);
    return eps;
}
__device__ __forceinline__ float band_eps(const ScanParamsH &p, float qn, float qres2, bool l2) {
    return band_eps_raw(p.xstat, p.d, qn, qres2, l2);
}

// exact score of the numeric contract for (query row, db row j); key to MAXIMISE
__device__ __forceinline__ float exact_score(const float *__restrict__ q, const float *__restrict__ x, int d,
                                             bool l2, float qn, float xn) {
    float acc = 0.0f;
    if ((d & 3) == 0) {
        const float4 *q4 = reinterpret_cast<const float4 *>(q);
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
#pragma unroll 8
        for (int c = 0; c < d / 4; ++c) {
            const float4 a = q4[c], b = x4[c];
            acc = __builtin_fmaf(a.x, b.x, acc);
            acc = __builtin_fmaf(a.y, b.y, acc);
            acc = __builtin_fmaf(a.z, b.z, acc);
            acc = __builtin_fmaf(a.w, b.w, acc);
        }
    } else {
        for (int c = 0; c < d; ++c) acc = __builtin_fmaf(q[c], x[c], acc);
    }
    if (l2) {
        const float dd = __builtin_fmaf(-2.0f, acc, qn + xn);
        return -(dd > 0.0f ? dd : 0.0f);
    }
    return acc;
}

// admission bound from tau (k-th largest approximate score): rows with s~ <= tau - 2 eps are out
__device__ __forceinline__ float bound_from_tau(float tau, float eps) {
    const float lo = tau - 2.0f * eps;
    return lo - fabsf(lo) * 2.4e-7f - 1e-37f;          // rounded DOWN
}

// "light" compaction of one query's approximate-key list (streaming kernel): no database access, no sort.
//   1. tau = k-th largest approximate score, by a 32-step bisection on the order-preserving score bits
//      (each step: 8 compares + 8 ballots per lane) -- O(32 n/64) instead of the O(n^2/64) rank-select;
//   2. every key that can still be in the exact top-k (s~ > tau - 2 eps) is stream-compacted to the
//      front of the list (ballot prefix sums; order does not matter until the final exact pass);
//   3. the admission bound is refreshed.  Returns the number of keys kept.
__device__ __forceinline__ int compact_light(u64 *__restrict__ list, int *cnt, float *thr_lo, float *thr_key,
                                             int row, int kk, float eps, int lane) {
    const int n = __builtin_amdgcn_readfirstlane(cnt[row]);
    if (n < kk) return n;
    u64 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (lane + 64 * i < n) ? list[lane + 64 * i] : 0;
    u32 o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (u32)(v[i] >> 32);
    u32 t = 0;                                          // largest t with #{ord >= t} >= kk  ==  kk-th largest ord
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const u32 cand = t | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) c += __builtin_popcountll(__ballot(o[i] >= cand));
        if (c >= kk) t = cand;                          // wave-uniform
    }
    const float lo = bound_from_tau(lemon_ord2f(t), eps);
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int base = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool keep = v[i] && lemon_key_score(v[i]) > lo;
        const u64 m = __ballot(keep);
        if (keep) list[base + __builtin_popcountll(m & below)] = v[i];
        base += __builtin_popcountll(m);
    }
    if (lane == 0) { cnt[row] = base; thr_lo[row] = lo; thr_key[row] = lo; }
    return base;
}

// exact compaction: re-score every entry of the list with the fp32 chain, keep the exact top-kk
// (sorted, exact keys).  Used once per query at the end of the scan, and when a band overflows.
__device__ __forceinline__ void compact_exact(const ScanParamsH &p, u64 *__restrict__ list, int *cnt, float *thr_lo,
                                              float *thr_key, int row, int64_t q, float eps, float qn, int lane,
                                              u64 *__restrict__ sk, u64 *__restrict__ sb) {
    const int n = __builtin_amdgcn_readfirstlane(cnt[row]);
    const int kk = p.b.kk;
    const float *qrow = p.q + q * (int64_t)p.d;
    const bool l2 = p.b.metric == LEMON_METRIC_L2;
    u64 best = 0;
#pragma unroll 1
    for (int base = 0; base < n; base += 64) {          // one exact chain per lane, then a 128-key rank merge
        const int e = base + lane;
        u64 key = 0;
        if (e < n) {
            const u32 j = lemon_key_index(list[e]);
            const float *xrow = p.x + (int64_t)j * p.d;
            const float s = exact_score(qrow, xrow, p.d, l2, qn, l2 ? p.b.xnorm[j] : 0.0f);
            key = (s == s) ? lemon_make_key(s, j) : 0;  // NaN is never selected (as in the fp32 scan)
        }
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
    const int have = __builtin_popcountll(__ballot(best != 0));
    const u64 kth = __shfl(best, kk - 1);
    if (lane == 0) {
        cnt[row] = have;
        if (have == kk) { const float lo = bound_from_tau(lemon_key_score(kth), eps); thr_lo[row] = lo; thr_key[row] = lo; }
    }
}

// post-tile maintenance of the streaming kernel for the 32 query rows of one wavefront.  Lane r inspects
// row r; only the rows that need work are visited.
__device__ __forceinline__ void maintain_rows(const ScanParamsH &p, u64 *__restrict__ cand_panel, int wave, int lane,
                                              int64_t q0, bool last, int *s_cnt, int *s_last, float *s_thr_lo,
                                              float *s_thr_key, const float *s_eps, const float *s_qn, u64 *sk, u64 *sb) {
    bool need = false;
    if (lane < 32) {
        const int row = 32 * wave + lane;
        const int c = s_cnt[row];
        const bool warm = c >= p.b.kk && s_thr_key[row] == -INFINITY;
        const bool stale = c >= p.b.kk && c - s_last[row] >= REFRESH;
        need = last ? (c > 0) : (c > CAPH - BX || warm || stale);
    }
    u64 todo = __ballot(need);
    while (todo) {
        const int r = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int row = 32 * wave + r;                  // wave-uniform (todo is a ballot)
        u64 *list = cand_panel + (int64_t)row * CAPH;
        const float eps = s_eps[row];
        if (!last) {
            const int kept = compact_light(list, s_cnt, s_thr_lo, s_thr_key, row, p.b.kk, eps, lane);
            if (lane == 0) s_last[row] = kept;
            if (kept > CAPH - BX)                        // the band itself does not fit: settle it exactly
                compact_exact(p, list, s_cnt, s_thr_lo, s_thr_key, row, q0 + row, eps, s_qn[row], lane, sk, sb);
        } else {
            compact_exact(p, list, s_cnt, s_thr_lo, s_thr_key, row, q0 + row, eps, s_qn[row], lane, sk, sb);
        }
    }
}

// ======================================================================================
// streaming variant (any d): both operands staged through LDS, like the fp32 scan
// ======================================================================================
template <bool L2>
__device__ __forceinline__ void epilogue_tile_h(f32x16 &acc, int rtile, u32 j, bool jvalid, float xn, int h,
                                                const float *s_thr_lo, const float *s_qn, int *s_cnt,
                                                u64 *__restrict__ cand_panel) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int rbase = rtile + 8 * g + 4 * h;
        const float4 t4 = *reinterpret_cast<const float4 *>(&s_thr_lo[rbase]);
        const float th[4] = {t4.x, t4.y, t4.z, t4.w};
        float qn[4] = {0.f, 0.f, 0.f, 0.f};
        if (L2) {
            const float4 n4 = *reinterpret_cast<const float4 *>(&s_qn[rbase]);
            qn[0] = n4.x; qn[1] = n4.y; qn[2] = n4.z; qn[3] = n4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = acc[4 * g + e];
            if (L2) {
                const float dd = __builtin_fmaf(-2.0f, s, qn[e] + xn);
                s = -(dd > 0.0f ? dd : 0.0f);
            }
            if (jvalid && s > th[e]) {
                const int row = rbase + e;
                const int slot = atomicAdd(&s_cnt[row], 1);     // (<= CAPH - 1: maintenance keeps c <= CAPH - BX before a tile)
                cand_panel[(int64_t)row * CAPH + (slot < CAPH ? slot : CAPH - 1)] = lemon_make_key(s, j);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
}

__global__ __launch_bounds__(NT, 2) void k_scan_bf16(ScanParamsH p) {
    __shared__ __attribute__((aligned(16))) float s_tile[2][2][BQ * BK];  // bf16 pairs: 128 rows x 64 bf16
    __shared__ __attribute__((aligned(16))) float s_thr_lo[BQ];
    __shared__ __attribute__((aligned(16))) float s_qn[BQ];
    __shared__ float s_thr_key[BQ];
    __shared__ float s_eps[BQ];
    __shared__ int s_cnt[BQ];
    __shared__ int s_last[BQ];
    __shared__ __attribute__((aligned(16))) u64 s_keys[NT / 64][256];
    __shared__ __attribute__((aligned(16))) u64 s_best[NT / 64][64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const int panel = blockIdx.x / p.b.splits;
    const int split = blockIdx.x % p.b.splits;
    const int64_t q0 = (int64_t)panel * BQ;
    const int t_begin = split * p.b.tiles_per_split;
    int t_end = t_begin + p.b.tiles_per_split;
    if (t_end > p.b.n_tiles) t_end = p.b.n_tiles;
    const int ntile = t_end - t_begin;
    const int KT = p.dpad_h / BKH;
    const int total = ntile * KT;
    const int dpad = p.dpad_h / 2;          // row pitch in 4-byte words (stage_ld counts floats)
    const unsigned voff = (unsigned)(((threadIdx.x >> 3) * dpad + 4 * (threadIdx.x & 7)) * 4);
    const int metric = p.b.metric;

    if (tid < BQ) {
        const bool valid = q0 + tid < p.b.nq;
        const float qn = p.b.qnorm[q0 + tid];
        s_eps[tid] = band_eps(p, qn, p.qres2[q0 + tid], metric == LEMON_METRIC_L2);
        s_qn[tid] = qn;
        s_thr_lo[tid] = valid ? -INFINITY : INFINITY;
        s_thr_key[tid] = valid ? -INFINITY : INFINITY;
        s_cnt[tid] = 0;
        s_last[tid] = 0;
    }

    float4 rq0, rq1, rq2, rq3, rx0, rx1, rx2, rx3;
    const float *qbase = reinterpret_cast<const float *>(p.qh + q0 * p.dpad_h);
    const float *xbase = reinterpret_cast<const float *>(p.xh + (int64_t)t_begin * BX * p.dpad_h);

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc00[e] = 0.0f; acc01[e] = 0.0f; acc10[e] = 0.0f; acc11[e] = 0.0f; }

    STAGE_ISSUE(qbase, xbase);
    STAGE_COMMIT(s_tile[0][0], s_tile[0][1]);
    __syncthreads();

    u64 *cand_panel = p.b.cand + (int64_t)blockIdx.x * BQ * CAPH;
    const int arow0 = 64 * wr + l31, arow1 = arow0 + 32;
    const int brow0 = 64 * wc + l31, brow1 = brow0 + 32;

    int kt = 0, jl = 0;
    for (int it = 0; it < total; ++it) {
        const int cur = it & 1;
        int kt_n = kt + 1, jl_n = jl;
        if (kt_n == KT) { kt_n = 0; jl_n = jl + 1; }
        if (it + 1 < total) {
            const float *qs = qbase + kt_n * BK;
            const float *xs = xbase + (int64_t)jl_n * BX * dpad + kt_n * BK;
            STAGE_ISSUE(qs, xs);
        }

        const float *tq = s_tile[cur][0];
        const float *tx = s_tile[cur][1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // 4 k-steps of 16
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(&tq[swz(arow0, 2 * u + h)]));
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(&tq[swz(arow1, 2 * u + h)]));
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(&tx[swz(brow0, 2 * u + h)]));
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(&tx[swz(brow1, 2 * u + h)]));
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc11, 0, 0, 0);
        }

        const bool tile_done = (kt == KT - 1);
        if (tile_done) {
            const int64_t jb = (int64_t)(t_begin + jl) * BX + 64 * wc + l31;
            const int64_t j0 = jb, j1 = jb + 32;
            const bool v0 = j0 < p.b.n, v1 = j1 < p.b.n;
            if (metric == LEMON_METRIC_L2) {
                const float xn0 = p.b.xnorm[j0], xn1 = p.b.xnorm[j1];
                epilogue_tile_h<true>(acc00, 64 * wr, (u32)j0, v0, xn0, h, s_thr_lo, s_qn, s_cnt, cand_panel);
                epilogue_tile_h<true>(acc01, 64 * wr, (u32)j1, v1, xn1, h, s_thr_lo, s_qn, s_cnt, cand_panel);
                epilogue_tile_h<true>(acc10, 64 * wr + 32, (u32)j0, v0, xn0, h, s_thr_lo, s_qn, s_cnt, cand_panel);
                epilogue_tile_h<true>(acc11, 64 * wr + 32, (u32)j1, v1, xn1, h, s_thr_lo, s_qn, s_cnt, cand_panel);
            } else {
                epilogue_tile_h<false>(acc00, 64 * wr, (u32)j0, v0, 0.f, h, s_thr_lo, s_qn, s_cnt, cand_panel);
                epilogue_tile_h<false>(acc01, 64 * wr, (u32)j1, v1, 0.f, h, s_thr_lo, s_qn, s_cnt, cand_panel);
                epilogue_tile_h<false>(acc10, 64 * wr + 32, (u32)j0, v0, 0.f, h, s_thr_lo, s_qn, s_cnt, cand_panel);
                epilogue_tile_h<false>(acc11, 64 * wr + 32, (u32)j1, v1, 0.f, h, s_thr_lo, s_qn, s_cnt, cand_panel);
            }
        }

        if (it + 1 < total) STAGE_COMMIT(s_tile[cur ^ 1][0], s_tile[cur ^ 1][1]);
        __syncthreads();

        if (tile_done) {
            maintain_rows(p, cand_panel, wave, lane, q0, it + 1 == total, s_cnt, s_last, s_thr_lo, s_thr_key, s_eps, s_qn,
                          s_keys[wave], s_best[wave]);
            __syncthreads();
        }
        kt = kt_n; jl = jl_n;
    }

    for (int r = 0; r < 32; ++r) {
        const int row = 32 * wave + r;
        const int64_t q = q0 + row;
        if (q >= p.b.nq) continue;
        const int kept = s_cnt[row];
        const u64 key = (lane < kept && lane < p.b.kk) ? cand_panel[(int64_t)row * CAPH + lane] : 0;
        write_out_row(p.b, split, q, lane, key);
    }
}

// ======================================================================================
// Q-stationary variant (d <= 1024): the query fragments never leave the registers.
//
// The streaming kernel above re-reads its 128-query panel from L2/MALL for every database tile
// (393 KB per workgroup at d=768: 64 panels per XCD thrash the 4 MB L2, PMC FETCH_SIZE ~ every staged
// byte).  Here a wavefront owns 32 queries and keeps their bf16 rows in VGPRs for the whole scan
// (16*KT x 4 VGPRs = 192 at d=768; one wave per SIMD, 512-register budget), as the B operand of
// v_mfma_f32_32x32x16_f16; only X tiles (128 rows x 64 k = 16 KB per stage) stream through LDS as
// the A operand, shared by the 4 waves.  Global traffic per flop halves, the L2 only sees the X
// stream that every workgroup reads in the same order, and the accumulator layout puts ONE query on
// each lane (col = lane&31), so the admission threshold is a per-lane scalar: a 16-value v_max3
// tree + one compare per accumulator tile instead of a compare+branch per element.
// ======================================================================================
// ---- Q-stationary candidate bookkeeping: everything lane-private --------------------------------
// A query is owned by the lane pair (l, l+32) of one wavefront; each of the two lanes appends to its
// OWN half of the query's list (256 entries each) with its count in a VGPR: an append is one
// predicated global store, no atomic, no LDS, no wait.  Thresholds live in VGPRs too.

// one 32x32 accumulator tile: a[e] = s~(db row jb + (e&3) + 8(e>>2), this lane's query)
template <bool l2>
__device__ __forceinline__ void qs_filter_tile(f32x16 a, float th, unsigned jb, float qn,
                                               const float *__restrict__ xnorm, unsigned n, int &ccnt,
                                               char *__restrict__ panel_bytes, unsigned my_off) {
    if (l2) {   // monotone proxy of the key -D: 2 s~ - |x|^2 = key + |q|^2 (th carries the same offset)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xn4 = *reinterpret_cast<const float4 *>(&xnorm[jb + 8 * g]);
            a[4 * g + 0] = __builtin_fmaf(2.0f, a[4 * g + 0], -xn4.x);
            a[4 * g + 1] = __builtin_fmaf(2.0f, a[4 * g + 1], -xn4.y);
            a[4 * g + 2] = __builtin_fmaf(2.0f, a[4 * g + 2], -xn4.z);
            a[4 * g + 3] = __builtin_fmaf(2.0f, a[4 * g + 3], -xn4.w);
        }
    }
    const float m0 = max3(max3(a[0], a[1], a[2]), a[3], a[3]), m1 = max3(max3(a[4], a[5], a[6]), a[7], a[7]);
    const float m2 = max3(max3(a[8], a[9], a[10]), a[11], a[11]), m3 = max3(max3(a[12], a[13], a[14]), a[15], a[15]);
    const float m = max3(max3(m0, m1, m2), m3, m3);      // 10 x v_max3_f32 (fmaxf: 19 instructions)
    // Survivors are rare (3-13 per wave and tile), and what the tests cost is not their instruction count but their
    // DEPENDENT hops: a per-lane `if` is v_cmp -> s_and_saveexec -> s_cbranch_execz, each waiting for the one before
    // (~17 such chains per entered tile in the nested per-lane form).  So the compares of a level are issued together as
    // wave ballots and each level is guarded by SCALAR branches on them: quads first, the four elements only of a quad
    // some lane passes in, the append itself under the per-lane test.  Three hops instead of seventeen.
    const u64 bq0 = __ballot(m0 > th), bq1 = __ballot(m1 > th), bq2 = __ballot(m2 > th), bq3 = __ballot(m3 > th);
    if ((bq0 | bq1 | bq2 | bq3) == 0) return;          // wave-uniform (m is what the quads' maxima fold to)
    (void)m;
    const u64 bq[4] = {bq0, bq1, bq2, bq3};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (bq[g]) {                                    // scalar branch
            u64 be[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) be[i] = __ballot(a[4 * g + i] > th);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (be[i]) {                            // scalar branch
                    const int e = 4 * g + i;
                    if (a[e] > th) {                    // (rows >= n: masked to -inf by the caller, last tile only)
                        // (the row number is made opaque here: otherwise hipcc strength-reduces the 64 `~j` key words of a
                        // tile into registers it carries around the whole scan loop)
                        unsigned j = jb + (e & 3) + 8 * (e >> 2);
                        asm volatile("" : "+v"(j));
                        // approximate key: for L2 the clamped -D~ = min(0, proxy - |q|^2)
                        const float s = l2 ? fminf(0.0f, a[e] - qn) : a[e];
                        *reinterpret_cast<u64 *>(panel_bytes + (my_off + 8u * append_slot(ccnt))) = lemon_make_key(s, j);
                        ++ccnt;
                    }
                }
            }
        }
    }
}

// entry idx (0..511) of a query's pair of half-lists holding n0 / n1 keys
__device__ __forceinline__ u64 qs_load_slot(const u64 *__restrict__ list, int idx, int n0, int n1) {
    const bool ok = idx < CAPH / 2 ? idx < n0 : (idx - CAPH / 2) < n1;
    return ok ? list[idx] : 0;
}

// light compaction of one query (both half-lists): bisection for the k-th largest approximate score,
// keep what can still matter, packed at the front of half-list 0 (then half-list 1).  Returns kept.
// The lists are read into NS register slots of 64 entries -- the occupied 64-blocks of half-list 0 first (s0 of them),
// then those of half-list 1; typical lists fill 2-3 of the 8 possible blocks, and every bisection step costs one
// compare + ballot + popcount per slot, so NS is specialised (as in the fp32 scan's PairSlots).
template <int NS>
__device__ __forceinline__ int qs_compact_light_ns(u64 *__restrict__ list, int n0, int n1, int s0, int kk, float eps, int lane,
                                                   float *lo_out) {
    u64 v[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const bool lo_half = i < s0;
        const int pos = lane + 64 * (lo_half ? i : i - s0);
        const bool ok = lo_half ? pos < n0 : pos < n1;
        v[i] = ok ? list[lo_half ? pos : CAPH / 2 + pos] : 0;
    }
    u32 o[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) o[i] = (u32)(v[i] >> 32);
    // t <= tau (the kk-th largest approximate score word): any lower bound of tau keeps the band a proof, so the
    // bisection stops at the first prefix that at most kk + 8 keys reach (~14 of the 32 steps)
    u32 t = 0;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const u32 cand = t | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) c += __builtin_popcountll(__ballot(o[i] >= cand));
        if (c >= kk) {
            t = cand;
            if (c <= kk + 8) break;
        }
    }
    const float lo = bound_from_tau(lemon_ord2f(t), eps);
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int base = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const bool keep = v[i] && lemon_key_score(v[i]) > lo;
        const u64 m = __ballot(keep);
        if (keep) list[base + __builtin_popcountll(m & below)] = v[i];   // contiguous over both halves (all loads precede)
        base += __builtin_popcountll(m);
    }
    *lo_out = lo;
    return base;
}
__device__ __forceinline__ int qs_compact_light(u64 *__restrict__ list, int n0, int n1, int kk, float eps, int lane,
                                                float *lo_out) {
    const int s0 = (n0 + 63) >> 6, ns = s0 + ((n1 + 63) >> 6);   // wave-uniform
    if (ns <= 2) return qs_compact_light_ns<2>(list, n0, n1, s0, kk, eps, lane, lo_out);
    if (ns == 3) return qs_compact_light_ns<3>(list, n0, n1, s0, kk, eps, lane, lo_out);
    if (ns == 4) return qs_compact_light_ns<4>(list, n0, n1, s0, kk, eps, lane, lo_out);
    return qs_compact_light_ns<8>(list, n0, n1, s0, kk, eps, lane, lo_out);
}

// exact compaction of one query: re-score every entry of both half-lists with the fp32 chain, leave the
// exact top-kk (sorted, exact keys) at the front of half-list 0.  Returns how many exist (<= kk).
__device__ __forceinline__ int qs_compact_exact(const ScanParamsH &p, u64 *__restrict__ list, int n0, int n1, int64_t q,
                                                float qn, int lane, u64 *__restrict__ sk, u64 *__restrict__ sb,
                                                u64 *kth_out) {
    const int kk = p.b.kk;
    const float *qrow = p.q + q * (int64_t)p.d;
    const bool l2 = p.b.metric == LEMON_METRIC_L2;
    u64 best = 0;
#pragma unroll 1
    for (int base = 0; base < CAPH; base += 64) {
        if (base < CAPH / 2 ? base >= n0 : (base - CAPH / 2) >= n1) continue;     // wave-uniform skip
        const u64 old = qs_load_slot(list, base + lane, n0, n1);
        u64 key = 0;
        if (old) {
            const u32 j = lemon_key_index(old);
            const float *xrow = p.x + (int64_t)j * p.d;
            const float s = exact_score(qrow, xrow, p.d, l2, qn, l2 ? p.b.xnorm[j] : 0.0f);
            key = (s == s) ? lemon_make_key(s, j) : 0;
        }
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
    *kth_out = __shfl(best, kk - 1);
    return __builtin_popcountll(__ballot(best != 0));
}

// LDS-DMA of one 16 KB X stage (128 rows x 128 B): each wave moves its 32 rows with four 1-KiB
// global_load_lds_dwordx4.  The LDS image is lane-linear (M0 base + 16*lane), so the bank swizzle
// of swz() is applied to the per-lane SOURCE chunk instead (guide rule 21).
__device__ __forceinline__ void qs_dma_stage(const float *__restrict__ src, int dpad, float *lds_stage, int wave,
                                             int lane) {
    const int rl = lane >> 3, c = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * wave + 8 * i + rl;
        const float *g = src + (int64_t)r * dpad + 4 * (c ^ ((r >> 1) & 7));
        float *l = lds_stage + (32 * wave + 8 * i) * BK;      // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)l, 16, 0, 0);
    }
}

// MFMA with the stationary operand (B = query fragment) and the accumulator in the ACCUMULATOR half
// of the unified register file: at d=768 the query fragments alone are 192 registers, which hipcc
// would otherwise keep in (and exhaust) the 256 architectural VGPRs, serialising every LDS read
// behind its MFMA.  "a" constraints pin both to AGPRs (64 acc + 192 query = all 256); the A
// fragments, addresses and the epilogue live in VGPRs.  First step of a tile uses C = 0.
__device__ __forceinline__ void mfma_qs_init(f32x16 &acc, bf16x8 a, const bf16x8 &bq) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "a"(bq));
}
__device__ __forceinline__ void mfma_qs(f32x16 &acc, bf16x8 a, const bf16x8 &bq) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(bq));
}

// (the metric is a template parameter: the L2 epilogue issues ordinary global loads (|x|^2), and hipcc
// drains the whole LDS-DMA queue before any use of an ordinary load while DMAs are in flight)
template <int KT, bool l2, bool PROF>
__global__ __launch_bounds__(NT, 1) void k_scan_bf16_qs(ScanParamsH p) {
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ts = 0;
#define PH_STAMP(acc) do { if (PROF) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - ts; ts = now_; } } while (0)
    constexpr int KS = 4 * KT;                 // 16-wide k steps
    constexpr int SUB = 2;                     // 64-wide k-slices per stage: 32 MFMAs per wave between barriers
    constexpr int KT2 = KT / SUB;              // stages per database tile
    constexpr int NB = 4;                      // LDS stage ring (4 x 32 KB); NB-1 stages of DMA in flight
    constexpr int LA = NB - 1;
    constexpr int STG = SUB * BX * BK;         // floats per stage
    // ONE shared array (with the DMA ring as its own object hipcc drains vmcnt before every k-step's
    // first ds_read: guide 5, trap (a)).  Candidate state is NOT in LDS: see the note above qs_filter_tile.
    __shared__ __attribute__((aligned(16))) float smem[NB * STG + 2 * BQ + (NT / 64) * (512 + 128)];
    float *s_x = smem;                                   // [NB][SUB][128*32]
    float *s_qn = smem + NB * STG;                       // [128]
    float *s_eps = s_qn + BQ;                            // [128]
    u64 *s_keys = reinterpret_cast<u64 *>(s_eps + BQ);   // [4][256] rank-select scratch
    u64 *s_best = s_keys + (NT / 64) * 256;              // [4][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    const int panel = blockIdx.x / p.b.splits;
    const int split = blockIdx.x % p.b.splits;
    const int64_t q0 = (int64_t)panel * BQ;
    int t_begin = split * p.b.tiles_per_split;
    int t_end = t_begin + p.b.tiles_per_split;
    if (t_end > p.b.n_tiles) t_end = p.b.n_tiles;
    if (p.b.splits == 1) { t_begin = p.chunk_t0; t_end = p.chunk_t1; }
    const int ntile = t_end - t_begin;
    const bool final_pass = (p.b.splits > 1) || p.last_chunk;
    const int dpad = p.dpad_h / 2;             // row pitch in 4-byte words

    if (tid < BQ) {
        const float qn = p.b.qnorm[q0 + tid];
        s_eps[tid] = band_eps(p, qn, p.qres2[q0 + tid], l2);
        s_qn[tid] = qn;
    }

    // ---- lane-private candidate state: query = 32*wave + (lane&31), half-list h = lane>>5 ----
    const int qrow_l = 32 * wave + l31;
    const bool qvalid = q0 + qrow_l < p.b.nq;
    const float my_qn = p.b.qnorm[q0 + qrow_l];
    u64 *cand_panel = p.b.cand + (int64_t)blockIdx.x * BQ * CAPH;
    char *panel_bytes = reinterpret_cast<char *>(cand_panel);          // appends: uniform base + 32-bit lane offset
    const unsigned my_off = (unsigned)(qrow_l * CAPH + h * (CAPH / 2)) * 8u;
    int ccnt = 0, clast = 0;                   // entries in my half-list; pair length right after the last compaction
    float thkey = qvalid ? -INFINITY : INFINITY;   // admission bound in key units (shared by the pair)
    if (p.b.splits == 1 && !p.first_chunk) {   // resume from the previous database chunk
        const float *st = p.state + 4 * ((int64_t)blockIdx.x * NT + tid);
        ccnt = __float_as_int(st[0]); clast = __float_as_int(st[1]); thkey = st[2];
    }
    auto th_of = [&](float tk) -> float {      // what the epilogue compares against (L2: proxy carries +|q|^2)
        if (!l2 || tk == -INFINITY || tk == INFINITY) return tk;
        return (tk + my_qn) - (fabsf(tk) + my_qn) * 2.4e-7f - 1e-37f;
    };
    float th = th_of(thkey);

    // ---- stationary operand: this lane's query row, all k ----
    bf16x8 qf[KS];
    {
        const lp16 *qsrc = p.qh + (q0 + qrow_l) * (int64_t)p.dpad_h + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8 *>(qsrc + 16 * s);
    }

    f32x16 acc0, acc1, acc2, acc3;
    // LDS byte address (inside slot 0, slice 0) of this lane's 16-B fragment chunk for each of the 4 chunk pairs of a
    // 64-wide k-slice; rows +32/+64/+96 and the second slice are immediate offsets (the swizzle term repeats every 16 rows)
    unsigned frag_addr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        frag_addr[u] = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_x + swz(l31, 2 * u + h));

    const float *xbase = reinterpret_cast<const float *>(p.xh + (int64_t)t_begin * BX * p.dpad_h);
    const int total = ntile * KT2;
    // stage s -> ring slot s & (NB-1); stage s covers k-slices SUB*(s % KT2) .. +SUB-1 of tile (s / KT2)
#define QS_ISSUE_STAGE(tile_base, kt2_, slot_)                                                         \
    do {                                                                                                \
        _Pragma("unroll") for (int sb_ = 0; sb_ < SUB; ++sb_)                                           \
            qs_dma_stage((tile_base) + (SUB * (kt2_) + sb_) * BK, dpad, s_x + (slot_) * STG + sb_ * BX * BK, wave, lane); \
    } while (0)
#pragma unroll
    for (int s0 = 0; s0 < LA; ++s0)
        if (s0 < total) QS_ISSUE_STAGE(xbase + (int64_t)(s0 / KT2) * BX * dpad, s0 % KT2, s0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (prologue only)
    __syncthreads();

    if (PROF) ts = __builtin_amdgcn_s_memtime();
    for (int jl = 0; jl < ntile; ++jl) {
        const float *xt = xbase + (int64_t)jl * BX * dpad;
#pragma clang loop unroll(full)
        for (int kt = 0; kt < KT2; ++kt) {
            const int t = jl * KT2 + kt;
            const bool more = t + LA < total;
            if (more) {   // stage t+LA into the slot stage t-1 was read from (everyone passed last barrier)
                const int kn = kt + LA;
                QS_ISSUE_STAGE(xt + (int64_t)(kn / KT2) * BX * dpad, kn % KT2, (t + LA) & (NB - 1));
            }
            // ---- the stage's 8 k-steps (SUB slices x 4 chunk pairs), software pipelined by hand ----
            // One wave per SIMD: nothing else hides the LDS latency, and hipcc schedules every ds_read directly in
            // front of the (inline-asm) MFMA that consumes it, with an lgkmcnt(0) in between -- 192 exposed LDS
            // round trips per tile, loop-only 0.42 of the bf16 peak.  So the fragment reads are issued explicitly:
            // two fragment sets, the reads of step s+1 go out before the four MFMAs of step s, and a counted
            // lgkmcnt(4) (the four newest reads may still be in flight) replaces the drain.
            {
                const unsigned sbase = (unsigned)((t & (NB - 1)) * STG * 4);
                const unsigned va0 = frag_addr[0] + sbase, va1 = frag_addr[1] + sbase, va2 = frag_addr[2] + sbase,
                               va3 = frag_addr[3] + sbase;
                bf16x8 fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3;
#define QS_LOAD(S, VA, OFF)                                                                                              \
                asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"                         \
                             "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"                             \
                             : "=&v"(f##S##0), "=&v"(f##S##1), "=&v"(f##S##2), "=&v"(f##S##3)                             \
                             : "v"(VA), "n"(OFF), "n"((OFF) + 4096), "n"((OFF) + 8192), "n"((OFF) + 12288) : "memory")
#define QS_WAIT(S, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f##S##0), "+v"(f##S##1), "+v"(f##S##2), "+v"(f##S##3))
#define QS_STEP(S, KSV)                                                                                                  \
                do {                                                                                                     \
                    const int ks_ = (KSV);                       /* compile-time after unrolling */                      \
                    if (ks_ == 0) {                                                                                      \
                        mfma_qs_init(acc0, f##S##0, qf[0]); mfma_qs_init(acc1, f##S##1, qf[0]);                          \
                        mfma_qs_init(acc2, f##S##2, qf[0]); mfma_qs_init(acc3, f##S##3, qf[0]);                          \
                    } else {                                                                                             \
                        mfma_qs(acc0, f##S##0, qf[ks_]); mfma_qs(acc1, f##S##1, qf[ks_]);                                \
                        mfma_qs(acc2, f##S##2, qf[ks_]); mfma_qs(acc3, f##S##3, qf[ks_]);                                \
                    }                                                                                                    \
                } while (0)
                static_assert(SUB == 2, "the k-step schedule below is written out for two 64-wide slices per stage");
                const int ks0 = 4 * SUB * kt;
                QS_LOAD(a, va0, 0);
                QS_LOAD(b, va1, 0);     QS_WAIT(a, 4); QS_STEP(a, ks0 + 0);
                QS_LOAD(a, va2, 0);     QS_WAIT(b, 4); QS_STEP(b, ks0 + 1);
                QS_LOAD(b, va3, 0);     QS_WAIT(a, 4); QS_STEP(a, ks0 + 2);
                QS_LOAD(a, va0, 16384); QS_WAIT(b, 4); QS_STEP(b, ks0 + 3);
                QS_LOAD(b, va1, 16384); QS_WAIT(a, 4); QS_STEP(a, ks0 + 4);
                QS_LOAD(a, va2, 16384); QS_WAIT(b, 4); QS_STEP(b, ks0 + 5);
                QS_LOAD(b, va3, 16384); QS_WAIT(a, 4); QS_STEP(a, ks0 + 6);
                                        QS_WAIT(b, 0); QS_STEP(b, ks0 + 7);
#undef QS_STEP
#undef QS_WAIT
#undef QS_LOAD
            }
            if (kt == KT2 - 1 && !(p.ablate & 1)) {
                PH_STAMP(ph0);
                // MFMA results are read by VALU next: hipcc pads nothing around asm, so wait out the
                // 16-pass MFMA latency here (once per tile)
                asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3));
                // ---- epilogue: acc[ni][e] = s~(db row 32ni + (e&3) + 8(e>>2) + 4h, query lane&31) ----
                const unsigned jb = (unsigned)(t_begin + jl) * BX + 4 * h;
                // (diagnostic, results invalid: LEMON_ABLATE bit 2 = nothing passes the filter, i.e. accumulator read-out
                // and maximum tree only: 97.0 ms at 262 144^2 x 768 against 91.2 with no epilogue at all and 145.8 whole --
                // the appends and compactions, not the read-out, are what the epilogue costs)
                if ((unsigned)(t_begin + jl + 1) * BX > (unsigned)p.b.n) {   // last tile of the database (uniform): padding rows never pass
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned j = jb + (e & 3) + 8 * (e >> 2);
                        if (j >= (unsigned)p.b.n) acc0[e] = -INFINITY;
                        if (j + 32 >= (unsigned)p.b.n) acc1[e] = -INFINITY;
                        if (j + 64 >= (unsigned)p.b.n) acc2[e] = -INFINITY;
                        if (j + 96 >= (unsigned)p.b.n) acc3[e] = -INFINITY;
                    }
                }
                const float th_f = (p.ablate & 4) ? INFINITY : th;
                qs_filter_tile<l2>(acc0, th_f, jb, my_qn, p.b.xnorm, (unsigned)p.b.n, ccnt, panel_bytes, my_off);
                qs_filter_tile<l2>(acc1, th_f, jb + 32, my_qn, p.b.xnorm, (unsigned)p.b.n, ccnt, panel_bytes, my_off);
                qs_filter_tile<l2>(acc2, th_f, jb + 64, my_qn, p.b.xnorm, (unsigned)p.b.n, ccnt, panel_bytes, my_off);
                qs_filter_tile<l2>(acc3, th_f, jb + 96, my_qn, p.b.xnorm, (unsigned)p.b.n, ccnt, panel_bytes, my_off);
                PH_STAMP(ph1);

                // ---- maintenance: which queries need a (light) compaction? ----
                const int pair = ccnt + __shfl_xor(ccnt, 32);
                const bool warm = thkey == -INFINITY && pair >= p.b.kk;
                const bool stale = pair >= p.b.kk && pair - clast >= p.b.stale;
                const bool full = ccnt > CAPH / 2 - BX / 2;           // my half could overflow on the next tile
                u64 todo = __ballot(qvalid && (warm || stale || full));
                todo = (todo | (todo >> 32)) & 0xffffffffull;          // one bit per query of this wave
                if (todo) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this tile's candidate stores must be visible
                    do {
                        const int r = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const int row = 32 * wave + r;
                        u64 *list = cand_panel + (int64_t)row * CAPH;
                        const int n0 = __builtin_amdgcn_readlane(ccnt, r), n1 = __builtin_amdgcn_readlane(ccnt, r + 32);
                        float lo;
                        int kept = qs_compact_light(list, n0, n1, p.b.kk, s_eps[row], lane, &lo);
                        if (kept > CAPH / 2 - BX / 2) {    // the band itself does not fit: settle it exactly
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            const int k0 = kept < CAPH / 2 ? kept : CAPH / 2;
                            u64 kth;
                            kept = qs_compact_exact(p, list, k0, kept - k0, q0 + row, s_qn[row], lane,
                                                    s_keys + wave * 256, s_best + wave * 64, &kth);
                            if (kept == p.b.kk) lo = bound_from_tau(lemon_key_score(kth), s_eps[row]);
                        }
                        if (l31 == r) {                    // both lanes of the pair take the new state
                            ccnt = h == 0 ? kept : 0;      // (kept <= CAPH/2 - BX/2 here: all in half-list 0)
                            clast = kept;
                            thkey = lo;
                            th = th_of(lo);
                        }
                    } while (todo);
                    PH_STAMP(ph3);
                }
            }
            // stage t+1 must have landed (all waves' parts) before anyone reads it: at most the LA-1
            // youngest stages (4 DMA instructions each) may still be in flight, then rendezvous
            if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // (LA-1) stages x 2 slices x 4 DMAs
            else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS reads retired
            if (PROF && (p.ablate & 8) && kt == KT2 - 1) PH_STAMP(ph3);   // diagnostic: the wait for the DMA counts as "maintain"
            __builtin_amdgcn_s_barrier();
            if (kt == KT2 - 1) PH_STAMP(ph2);
        }
    }

    if (PROF && tid == 0) {
        atomicAdd(&p.phase_dbg[0], ph0); atomicAdd(&p.phase_dbg[1], ph1);
        atomicAdd(&p.phase_dbg[2], ph2); atomicAdd(&p.phase_dbg[3], ph3);
    }
#undef PH_STAMP
#undef QS_ISSUE_STAGE
    if (!final_pass) {      // park the lane-private state for the next database chunk
        float *st = p.state + 4 * ((int64_t)blockIdx.x * NT + tid);
        st[0] = __int_as_float(ccnt); st[1] = __int_as_float(clast); st[2] = thkey;
        return;
    }
    // ---- end of the scan: the exact re-scoring + exact top-k of every query runs in its own kernel (k_bf16_final: one
    // wave per query, many waves per CU); here only the half-list counts are handed over ----
    p.cnt[2 * ((int64_t)blockIdx.x * BQ + qrow_l) + h] = ccnt;
}

// ======================================================================================
// Q-stationary variant with TWO 32-query blocks per wave ("QS2").
//
// What bounded k_scan_bf16_qs's loop (tools/micro/qs_loop.hip, qs2_loop.hip; same box, random operands, all workgroups
// walking one chunk in step): the delivery of the database tile.  The barrier + LDS-DMA skeleton WITHOUT any fragment
// read or dependent MFMA wait already stops at 0.55 of the bf16 peak (full clock, zero or random data alike): a workgroup of
// 128 queries needs a 16 KB slice per 16 MFMAs of a wave = 32 B/clk/CU at the MFMA peak, which the L2 -> LDS path does not
// sustain, and every MFMA reads 1 KB of LDS on top.  Here a wave owns 64 queries: block 0 in the 192 AccVGPRs next to the
// 64 accumulator registers (as before), block 1 in ARCHITECTURAL VGPRs -- the MFMA B operand may come from either half of
// the unified file -- except its last PARK k-steps, which are parked in LDS (lane-linear, conflict-free ds_read_b128)
// because 192 + ~90 working registers do not fit 256.  A database tile is 64 rows, every A-fragment read feeds two MFMAs,
// a workgroup covers 256 queries: LDS bytes per MFMA and global -> LDS bytes per flop both halve at the same 32 MFMAs per
// wave between barriers.  Loop replica: 1 269 -> 1 534 TFLOP/s (0.508 -> 0.614 of 2.5 PF; 0.635 -> 0.812 on all-zero data).
// Candidate bookkeeping is the QS kernel's, per block: thresholds / counts in VGPRs, lane-private half-lists.
// Tiles, splits and chunks are counted in 64-ROW units here (p.b.n_tiles, tiles_per_split, chunk_t0/t1).
// ======================================================================================
constexpr int RT2 = 64;        // database rows per tile
constexpr int BQ2 = 256;       // queries per workgroup (64 per wave)

// one MFMA with the accumulator tile in the AccVGPR (CV = false) or the architectural (CV = true) half of the register file
// and the stationary B operand in AccVGPRs (BA) or VGPRs; INIT: C = 0
template <bool CV, bool BA, bool INIT>
__device__ __forceinline__ void mfma_x(f32x16 &acc, bf16x8 a, const bf16x8 &bq) {
    if (INIT) {
        // (early-clobber: a multi-pass MFMA may write its destination before it has read all of A / B, so the fresh tile
        // must not share registers with the fragments -- hipcc would otherwise reuse a dying fragment's VGPRs for it)
        if (CV) { if (BA) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(bq));
                  else    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(bq)); }
        else    { if (BA) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "a"(bq));
                  else    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "v"(bq)); }
    } else {
        if (CV) { if (BA) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(bq));
                  else    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(bq)); }
        else    { if (BA) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(bq));
                  else    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(bq)); }
    }
}

// LDS-DMA of one 64-row x 128-B slice (8 KB): each wave moves 16 rows with two 1-KiB global_load_lds_dwordx4, issued by hand
// as `SGPR base + 32-bit lane offset` (the builtin wants a 64-bit per-lane pointer: two VGPRs and a v_lshl_add_u64 per DMA;
// the lane offsets below are constants of the kernel).  M0 = LDS destination of the wave's 1 KiB; nothing else in this
// kernel uses M0 (no builtin DMA, no movrel), so it is not declared clobbered (hipcc warns that it is reserved).
__device__ __forceinline__ void qs2_dma_slice(const float *__restrict__ src, unsigned lds_bytes, unsigned voff0, unsigned voff1) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff0), "s"(src), "s"(lds_bytes) : "memory");
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff1), "s"(src), "s"(lds_bytes + 1024u) : "memory");
}

// ACCV: the four accumulator tiles live in ARCHITECTURAL VGPRs (the epilogue then reads them with VALU instructions directly:
// no 64 v_accvgpr_read per tile) and the first NA k-steps of query block 1 take the 64 AccVGPRs they leave free
template <int KT, int PARK, bool l2, bool PROF, bool ACCV>
__global__ __launch_bounds__(NT, 1) void k_scan_bf16_qs2(ScanParamsH p) {
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ts = 0;
#define PH_STAMP(acc) do { if (PROF) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - ts; ts = now_; } } while (0)
    constexpr int KS = 4 * KT;                 // 16-wide k steps
    constexpr int NR = KS - PARK;              // block-1 fragments kept in registers
    constexpr int NA = ACCV ? (NR < 16 ? NR : 16) : 0;   // ... of which in AccVGPRs
    constexpr int SUB = 2, KT2 = KT / SUB, NB = 4, LA = NB - 1;
    constexpr int STG = SUB * RT2 * BK;        // floats per stage (16 KB)
    constexpr int PK = PARK ? PARK : 1;
    static_assert(NT == BQ2 && NR > 0 && KT % SUB == 0, "one thread per query row in the prologue");
    __shared__ __attribute__((aligned(16))) float smem[NB * STG + (NT / 64) * PK * 256 + 2 * BQ2 + (NT / 64) * (512 + 128)];
    float *s_x = smem;                                   // [NB][SUB][64 * 32]
    float *s_q = smem + NB * STG;                        // [4 waves][PARK][64 lanes x 16 B] parked block-1 fragments
    float *s_qn = s_q + (NT / 64) * PK * 256;            // [256]
    float *s_eps = s_qn + BQ2;                           // [256]
    u64 *s_keys = reinterpret_cast<u64 *>(s_eps + BQ2);  // [4][256] rank-select scratch
    u64 *s_best = s_keys + (NT / 64) * 256;              // [4][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    const int panel = blockIdx.x / p.b.splits;
    const int split = blockIdx.x % p.b.splits;
    const int64_t q0 = (int64_t)panel * BQ2;
    int t_begin = split * p.b.tiles_per_split;
    int t_end = t_begin + p.b.tiles_per_split;
    if (t_end > p.b.n_tiles) t_end = p.b.n_tiles;
    if (p.b.splits == 1) { t_begin = p.chunk_t0; t_end = p.chunk_t1; }
    const int ntile = t_end - t_begin;
    const bool final_pass = (p.b.splits > 1) || p.last_chunk;
    const int dpad = p.dpad_h / 2;             // row pitch in 4-byte words

    {
        const float qn = p.b.qnorm[q0 + tid];
        s_eps[tid] = band_eps(p, qn, p.qres2[q0 + tid], l2);
        s_qn[tid] = qn;
    }

    // ---- lane-private candidate state, per query block b: query = 64*wave + 32*b + (lane&31), half-list h = lane>>5 ----
    const int qrow0 = 64 * wave + l31, qrow1 = qrow0 + 32;
    const bool qvalid0 = q0 + qrow0 < p.b.nq, qvalid1 = q0 + qrow1 < p.b.nq;
    const float my_qn0 = p.b.qnorm[q0 + qrow0], my_qn1 = p.b.qnorm[q0 + qrow1];
    u64 *cand_panel = p.b.cand + (int64_t)blockIdx.x * BQ2 * CAPH;
    char *panel_bytes = reinterpret_cast<char *>(cand_panel);
    const unsigned my_off0 = (unsigned)(qrow0 * CAPH + h * (CAPH / 2)) * 8u, my_off1 = (unsigned)(qrow1 * CAPH + h * (CAPH / 2)) * 8u;
    int ccnt0 = 0, clast0 = 0, ccnt1 = 0, clast1 = 0;
    float thkey0 = qvalid0 ? -INFINITY : INFINITY, thkey1 = qvalid1 ? -INFINITY : INFINITY;
    if (p.b.splits == 1 && !p.first_chunk) {   // resume from the previous database chunk
        const float *st = p.state + 8 * ((int64_t)blockIdx.x * NT + tid);
        ccnt0 = __float_as_int(st[0]); clast0 = __float_as_int(st[1]); thkey0 = st[2];
        ccnt1 = __float_as_int(st[4]); clast1 = __float_as_int(st[5]); thkey1 = st[6];
    }
    auto th_of = [&](float tk, float qn) -> float {      // what the epilogue compares against (L2: proxy carries +|q|^2)
        if (!l2 || tk == -INFINITY || tk == INFINITY) return tk;
        return (tk + qn) - (fabsf(tk) + qn) * 2.4e-7f - 1e-37f;
    };
    float th0 = th_of(thkey0, my_qn0), th1 = th_of(thkey1, my_qn1);

    // ---- stationary operands: this lane's two query rows ----
    bf16x8 qa[KS], qb[NR];
    {
        const lp16 *src0 = p.qh + (q0 + qrow0) * (int64_t)p.dpad_h + 8 * h;
        const lp16 *src1 = p.qh + (q0 + qrow1) * (int64_t)p.dpad_h + 8 * h;
        // (loads in groups of eight with a scheduling fence in between: left alone hipcc issues all 84 loads up front,
        // needs 336 registers for them and spills the query rows through scratch on their way into the AccVGPRs)
#pragma unroll
        for (int s = NR; s < KS; ++s)          // (prologue: before any LDS-DMA is in flight)
            *reinterpret_cast<bf16x8 *>(s_q + ((wave * PK + (s - NR)) * 64 + lane) * 4) = *reinterpret_cast<const bf16x8 *>(src1 + 16 * s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qa[s] = *reinterpret_cast<const bf16x8 *>(src0 + 16 * s);
            if ((s & 7) == 7) { asm volatile("" : "+a"(qa[s])); __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int s = 0; s < NR; ++s) {
            qb[s] = *reinterpret_cast<const bf16x8 *>(src1 + 16 * s);
            if ((s & 7) == 7) { if (s < NA) asm volatile("" : "+a"(qb[s])); else asm volatile("" : "+v"(qb[s])); __builtin_amdgcn_sched_barrier(0); }
        }
    }

    f32x16 acc00, acc01, acc10, acc11;         // acc<block><row half>
    unsigned frag_addr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        frag_addr[u] = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_x + swz(l31, 2 * u + h));
    const unsigned vq = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_q + (wave * PK * 64 + lane) * 4);

    const float *xbase = reinterpret_cast<const float *>(p.xh + (int64_t)t_begin * RT2 * p.dpad_h);
    const int total = ntile * KT2;
    // a wave moves rows 16 wave + 8 i + lane/8 of a slice (i = 0, 1), 16-B chunk (lane&7) ^ swizzle(row) of each
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)s_x + (unsigned)wave * 2048u;
    const unsigned voff0 = (unsigned)(((16 * wave + (lane >> 3)) * dpad + 4 * ((lane & 7) ^ ((lane >> 4) & 7))) * 4);
    const unsigned voff1 = (unsigned)(((16 * wave + 8 + (lane >> 3)) * dpad + 4 * ((lane & 7) ^ ((4 + (lane >> 4)) & 7))) * 4);
#define Q2_ISSUE_STAGE(tile_base, kt2_, slot_)                                                          \
    do {                                                                                                \
        _Pragma("unroll") for (int sb_ = 0; sb_ < SUB; ++sb_)                                           \
            qs2_dma_slice((tile_base) + (SUB * (kt2_) + sb_) * BK, lds0 + (unsigned)(((slot_) * STG + sb_ * RT2 * BK) * 4), voff0, voff1); \
    } while (0)
#pragma unroll
    for (int s0 = 0; s0 < LA; ++s0)
        if (s0 < total) Q2_ISSUE_STAGE(xbase + (int64_t)(s0 / KT2) * RT2 * dpad, s0 % KT2, s0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (prologue only)
    __syncthreads();

    if (PROF) ts = __builtin_amdgcn_s_memtime();
    for (int jl = 0; jl < ntile; ++jl) {
        const float *xt = xbase + (int64_t)jl * RT2 * dpad;
#pragma clang loop unroll(full)
        for (int kt = 0; kt < KT2; ++kt) {
            const int t = jl * KT2 + kt;
            const bool more = t + LA < total;
            if (more) {   // stage t+LA into the slot stage t-1 was read from (everyone passed the last barrier)
                const int kn = kt + LA;
                Q2_ISSUE_STAGE(xt + (int64_t)(kn / KT2) * RT2 * dpad, kn % KT2, (t + LA) & (NB - 1));
            }
            // ---- the stage's 8 k-steps: fragment reads one k-step ahead, two fragment sets, counted lgkmcnt ----
            {
                const unsigned sbase = (unsigned)((t & (NB - 1)) * STG * 4);
                const int ks0 = 4 * SUB * kt;
                bf16x8 fa0, fa1, fb0, fb1, pa, pb;      // p<set>: the parked block-1 fragment of the step, when it has one
#define Q2_LOADS(S, KSV)                                                                                                 \
                do {                                                                                                     \
                    const unsigned va_ = frag_addr[(KSV) & 3] + sbase + (((KSV) & 4) ? (unsigned)(RT2 * 128) : 0u);      \
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"                                \
                                 : "=&v"(f##S##0), "=&v"(f##S##1) : "v"(va_) : "memory");                                \
                    if (PARK && ks0 + (KSV) >= NR) {                                                                     \
                        const unsigned vp_ = vq + (unsigned)((ks0 + (KSV) - NR) * 1024);                                 \
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(p##S) : "v"(vp_) : "memory");                         \
                    }                                                                                                    \
                } while (0)
#define Q2_NRD(KSV) ((KSV) > 7 ? 0 : ((PARK && ks0 + (KSV) >= NR) ? 3 : 2))     /* reads the step's LOADS issues */
#define Q2_WAIT(S, N)                                                                                                    \
                do {                                                                                                     \
                    if ((N) == 3)      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(p##S)); \
                    else if ((N) == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(p##S)); \
                    else               asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(p##S)); \
                } while (0)
#define Q2_STEP(S, KSV)                                                                                                  \
                do {                                                                                                     \
                    const int ks_ = ks0 + (KSV);                     /* compile-time after unrolling */                  \
                    if (ks_ == 0) {                                                                                      \
                        mfma_x<ACCV, true, true>(acc00, f##S##0, qa[0]); mfma_x<ACCV, true, true>(acc01, f##S##1, qa[0]); \
                        mfma_x<ACCV, (NA > 0), true>(acc10, f##S##0, qb[0]); mfma_x<ACCV, (NA > 0), true>(acc11, f##S##1, qb[0]); \
                    } else {                                                                                             \
                        mfma_x<ACCV, true, false>(acc00, f##S##0, qa[ks_]); mfma_x<ACCV, true, false>(acc01, f##S##1, qa[ks_]); \
                        if (ks_ < NA)      { mfma_x<ACCV, true, false>(acc10, f##S##0, qb[ks_ < NR ? ks_ : 0]); mfma_x<ACCV, true, false>(acc11, f##S##1, qb[ks_ < NR ? ks_ : 0]); } \
                        else if (ks_ < NR) { mfma_x<ACCV, false, false>(acc10, f##S##0, qb[ks_ < NR ? ks_ : 0]); mfma_x<ACCV, false, false>(acc11, f##S##1, qb[ks_ < NR ? ks_ : 0]); } \
                        else               { mfma_x<ACCV, false, false>(acc10, f##S##0, p##S); mfma_x<ACCV, false, false>(acc11, f##S##1, p##S); } \
                    }                                                                                                    \
                } while (0)
                // (fragment reads TWO k-steps ahead over three sets were measured too: 1 404.5 vs 1 400.6 ms at 1 M x 768 -- nothing)
                pa = fa0 = fa1 = pb = fb0 = fb1 = bf16x8{};
                Q2_LOADS(a, 0);
                Q2_LOADS(b, 1); Q2_WAIT(a, Q2_NRD(1)); Q2_STEP(a, 0);
                Q2_LOADS(a, 2); Q2_WAIT(b, Q2_NRD(2)); Q2_STEP(b, 1);
                Q2_LOADS(b, 3); Q2_WAIT(a, Q2_NRD(3)); Q2_STEP(a, 2);
                Q2_LOADS(a, 4); Q2_WAIT(b, Q2_NRD(4)); Q2_STEP(b, 3);
                Q2_LOADS(b, 5); Q2_WAIT(a, Q2_NRD(5)); Q2_STEP(a, 4);
                Q2_LOADS(a, 6); Q2_WAIT(b, Q2_NRD(6)); Q2_STEP(b, 5);
                Q2_LOADS(b, 7); Q2_WAIT(a, Q2_NRD(7)); Q2_STEP(a, 6);
                                Q2_WAIT(b, 0);         Q2_STEP(b, 7);
#undef Q2_STEP
#undef Q2_WAIT
#undef Q2_NRD
#undef Q2_LOADS
            }
            if (kt == KT2 - 1 && !(p.ablate & 1)) {
                PH_STAMP(ph0);
                if ((unsigned)(t_begin + jl + 1) * RT2 > (unsigned)p.b.n) {
                    // last tile of the database (uniform): its padding rows must never pass.  One more MFMA per accumulator
                    // tile does it: A' = -inf in k-slot 0 of the padding rows (0 elsewhere), B' = 1 in k-slot 0 (0
                    // elsewhere), so every score of a padding row becomes -inf and a valid row's gets +0.  (Masking the 64
                    // accumulator values with VALU code costs 64 live VGPRs + 32 compare masks that this kernel does not have.)
                    const unsigned row = (unsigned)(t_begin + jl) * RT2 + (unsigned)l31;
                    const unsigned short ninf = 0xfc00u, one = 0x3c00u;       // fp16 -inf, 1.0
                    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                    u16x8 m0 = {}, m1 = {}, ones = {};
                    m0[0] = (h == 0 && row >= (unsigned)p.b.n) ? ninf : (unsigned short)0;
                    m1[0] = (h == 0 && row + 32 >= (unsigned)p.b.n) ? ninf : (unsigned short)0;
                    ones[0] = h == 0 ? one : (unsigned short)0;
                    const bf16x8 a0 = __builtin_bit_cast(bf16x8, m0), a1 = __builtin_bit_cast(bf16x8, m1), b1 = __builtin_bit_cast(bf16x8, ones);
                    mfma_x<ACCV, false, false>(acc00, a0, b1); mfma_x<ACCV, false, false>(acc01, a1, b1);
                    mfma_x<ACCV, false, false>(acc10, a0, b1); mfma_x<ACCV, false, false>(acc11, a1, b1);
                }
                if (ACCV) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc00), "+v"(acc01), "+v"(acc10), "+v"(acc11));
                else      asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc00), "+a"(acc01), "+a"(acc10), "+a"(acc11));
                // ---- epilogue: acc<b><i>[e] = s~(db row 32 i + (e&3) + 8(e>>2) + 4h of the tile, query block b's lane&31) ----
                const unsigned jb = (unsigned)(t_begin + jl) * RT2 + 4 * h;
                const float tf0 = (p.ablate & 4) ? INFINITY : th0, tf1 = (p.ablate & 4) ? INFINITY : th1;
                // (L2: scheduling fences keep hipcc from hoisting all sixteen |x|^2 float4 loads of the four calls to the top)
                qs_filter_tile<l2>(acc00, tf0, jb, my_qn0, p.b.xnorm, (unsigned)p.b.n, ccnt0, panel_bytes, my_off0);
                if (l2) __builtin_amdgcn_sched_barrier(0);
                qs_filter_tile<l2>(acc01, tf0, jb + 32, my_qn0, p.b.xnorm, (unsigned)p.b.n, ccnt0, panel_bytes, my_off0);
                if (l2) __builtin_amdgcn_sched_barrier(0);
                qs_filter_tile<l2>(acc10, tf1, jb, my_qn1, p.b.xnorm, (unsigned)p.b.n, ccnt1, panel_bytes, my_off1);
                if (l2) __builtin_amdgcn_sched_barrier(0);
                qs_filter_tile<l2>(acc11, tf1, jb + 32, my_qn1, p.b.xnorm, (unsigned)p.b.n, ccnt1, panel_bytes, my_off1);
                PH_STAMP(ph1);

                // ---- maintenance, per query block: which queries need a (light) compaction? ----
                bool waited = false;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int &ccnt = b ? ccnt1 : ccnt0, &clast = b ? clast1 : clast0;
                    float &thkey = b ? thkey1 : thkey0, &th = b ? th1 : th0;
                    const float my_qn = b ? my_qn1 : my_qn0;
                    const bool qvalid = b ? qvalid1 : qvalid0;
                    const int pair = ccnt + __shfl_xor(ccnt, 32);
                    const bool warm = thkey == -INFINITY && pair >= p.b.kk;
                    const bool stale = pair >= p.b.kk && pair - clast >= p.b.stale;
                    const bool full = ccnt > CAPH / 2 - RT2 / 2;          // my half could overflow on the next tile (<= 32 appends)
                    u64 todo = __ballot(qvalid && (warm || stale || full));
                    todo = (todo | (todo >> 32)) & 0xffffffffull;          // one bit per query of this block
                    if (!todo) continue;
                    if (!waited) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); waited = true; }   // this tile's appends are visible
                    do {
                        const int r = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const int row = 64 * wave + 32 * b + r;
                        u64 *list = cand_panel + (int64_t)row * CAPH;
                        const int n0 = __builtin_amdgcn_readlane(ccnt, r), n1 = __builtin_amdgcn_readlane(ccnt, r + 32);
                        float lo;
                        int kept = qs_compact_light(list, n0, n1, p.b.kk, s_eps[row], lane, &lo);
                        if (kept > CAPH / 2 - RT2 / 2) {    // the band itself does not fit: settle it exactly
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            const int k0 = kept < CAPH / 2 ? kept : CAPH / 2;
                            u64 kth;
                            kept = qs_compact_exact(p, list, k0, kept - k0, q0 + row, s_qn[row], lane,
                                                    s_keys + wave * 256, s_best + wave * 64, &kth);
                            if (kept == p.b.kk) lo = bound_from_tau(lemon_key_score(kth), s_eps[row]);
                        }
                        if (l31 == r) {                    // both lanes of the pair take the new state
                            ccnt = h == 0 ? kept : 0;      // (kept <= CAPH/2 - 32 here: all in half-list 0)
                            clast = kept;
                            thkey = lo;
                            th = th_of(lo, my_qn);
                        }
                    } while (todo);
                }
                if (waited) PH_STAMP(ph3);
            }
            // stage t+1 must have landed (all waves' parts) before anyone reads it: at most the LA-1 youngest stages
            // (4 DMA instructions each; younger appends only make the wait longer, never shorter) may still be in flight
            if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS reads retired
            if (PROF && (p.ablate & 8) && kt == KT2 - 1) PH_STAMP(ph3);
            __builtin_amdgcn_s_barrier();
            if (kt == KT2 - 1) PH_STAMP(ph2);
        }
    }

    if (PROF && tid == 0) {
        atomicAdd(&p.phase_dbg[0], ph0); atomicAdd(&p.phase_dbg[1], ph1);
        atomicAdd(&p.phase_dbg[2], ph2); atomicAdd(&p.phase_dbg[3], ph3);
    }
#undef PH_STAMP
#undef Q2_ISSUE_STAGE
    if (!final_pass) {      // park the lane-private state for the next database chunk
        float *st = p.state + 8 * ((int64_t)blockIdx.x * NT + tid);
        st[0] = __int_as_float(ccnt0); st[1] = __int_as_float(clast0); st[2] = thkey0;
        st[4] = __int_as_float(ccnt1); st[5] = __int_as_float(clast1); st[6] = thkey1;
        return;
    }
    // ---- end of the scan: counts for k_bf16_final (exact re-scoring + exact top-k, one wave per query) ----
    p.cnt[2 * ((int64_t)blockIdx.x * BQ2 + qrow0) + h] = ccnt0;
    p.cnt[2 * ((int64_t)blockIdx.x * BQ2 + qrow1) + h] = ccnt1;
}

// ======================================================================================
// QS4 (round 5): the Q-stationary scan on v_mfma_f32_16x16x32_f16.
//
// Same work per workgroup as QS2 (256 queries, 64 per wave, 64-row database tiles, 4 x 16 KB LDS ring fed by LDS-DMA, one
// wave per SIMD, 512 registers) and the same LDS bytes per flop, but the wave's 64 x 64 tile is 4 x 4 accumulator tiles of
// 16 x 16 (4 registers each): per k32 step 4 database fragments (16 rows x 32 k) x 4 query fragments (16 queries x 32 k) =
// 16 MFMAs of 16 cycles.  Why (tools/micro/qs3_loop.hip against qs2_loop.hip on one box, random operands, all workgroups
// in step): the 16x16x32 form draws less power per flop -- the board holds 2.14 GHz instead of 1.89 -- and its issue
// granularity lets everything else ride BETWEEN the MFMAs instead of in front of them:
//   * fragment reads of step s + 1 and the stage's four DMA pieces are issued one per MFMA gap of step s (an MFMA holds
//     the vector issue for 8 of its 16 cycles): loop replica 1 431 (QS2) -> 1 481 (same schedule) -> 1 527 (DMA spread)
//     -> 1 565 TFLOP/s (reads interleaved too) on the slower of two boxes, 1 500 -> 1 615 on the faster;
//   * the ring's barrier sits in the MIDDLE of a stage (behind step 1 of 4): stage t + 1 has landed for every wave two
//     steps before anybody needs it, so the first fragments of a stage are read during the last step of the one before
//     (QS2 reads them behind the barrier: one exposed LDS round trip per stage), and the slot of stage t - 1 is free for
//     the DMA of stage t + 3 in steps 2 and 3;
//   * the filter of tile j runs between the MFMAs of tile j + 1's FIRST step: that step is ordered by query group
//     (b-major), the MFMAs of group b start from C = 0, and the filter of group b -- eight v_max3 over the lane's sixteen
//     scores of that group, one compare, one scalar branch -- sits in front of them while the MFMAs of group b - 1 run.
//     No accumulator read-out phase, no s_nop: the values a filter reads were finished twelve MFMAs earlier.
// Accumulator layout C[row = 4 (lane >> 4) + i][col = lane & 15]: a QUERY lives on the four lanes c, c + 16, c + 32, c + 48,
// each of which sees 16 of a tile's 64 rows; every lane appends to its own QUARTER of the query's list (128 entries, count
// in a VGPR).  A light compaction reads the four quarters as one list and deals the survivors back round-robin, so the
// quarters stay balanced and `full` (a quarter within one tile of its capacity) practically never fires.
// Fragment homes: (group b, step s) -> index f = b NS + s; f < 64 in AccVGPRs, the rest in VGPRs, group 3's last PARK
// steps in LDS (lane-linear).  Tiles, splits and chunks are counted in 64-ROW units, as in QS2.
// ======================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int QCAP4 = CAPH / 4;    // entries per lane list (quarter of a query's list)

template <bool BA, bool INIT>
__device__ __forceinline__ void mfma16(f32x4 &acc, bf16x8 a, const bf16x8 &bq) {
    // (the accumulator is "+v" in the C = 0 form too: the filter's reads of the finished tile must stay in front of it, and the
    // new tile must take the old one's registers)
    if (INIT) { if (BA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+v"(acc) : "v"(a), "a"(bq));
                else    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+v"(acc) : "v"(a), "v"(bq)); }
    else      { if (BA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(bq));
                else    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(bq)); }
}

// one 1-KiB piece of a 64-row x 128-B slice: the wave's rows 16 wave + 8 half + lane / 8 (see qs2_dma_slice)
__device__ __forceinline__ void qs4_dma_piece(const float *__restrict__ src, unsigned lds_bytes, unsigned voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src), "s"(lds_bytes) : "memory");
}

// the same with the source's constant part in the instruction's offset field (one SGPR pair serves the four pieces of a stage)
template <int SRC_OFF>
__device__ __forceinline__ void qs4_dma_piece_off(const float *__restrict__ src, unsigned lds_bytes, unsigned voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(voff), "s"(src), "s"(lds_bytes), "n"(SRC_OFF) : "memory");
}

__device__ __forceinline__ unsigned append_slot4(int cnt) {      // (see append_slot: a broken invariant must not leave the list)
    return (unsigned)cnt < (unsigned)(QCAP4 - 1) ? (unsigned)cnt : (unsigned)(QCAP4 - 1);
}

// the lane's sixteen scores of one query group: c<a>[i] = s~(row jb + 16 a + i, the lane's query of the group).
// qs4_group_max: (L2: the scores become the monotone proxy of the key in place, then) their maximum -- eight v_max3_f32.
template <bool l2>
__device__ __forceinline__ float qs4_group_max(f32x4 &c0, f32x4 &c1, f32x4 &c2, f32x4 &c3, unsigned jb, const float *__restrict__ xnorm) {
    if (l2) {   // monotone proxy of the key -D: 2 s~ - |x|^2 = key + |q|^2 (th carries the same offset)
        const float4 x0 = *reinterpret_cast<const float4 *>(&xnorm[jb]), x1 = *reinterpret_cast<const float4 *>(&xnorm[jb + 16]);
        const float4 x2 = *reinterpret_cast<const float4 *>(&xnorm[jb + 32]), x3 = *reinterpret_cast<const float4 *>(&xnorm[jb + 48]);
        c0[0] = __builtin_fmaf(2.0f, c0[0], -x0.x); c0[1] = __builtin_fmaf(2.0f, c0[1], -x0.y); c0[2] = __builtin_fmaf(2.0f, c0[2], -x0.z); c0[3] = __builtin_fmaf(2.0f, c0[3], -x0.w);
        c1[0] = __builtin_fmaf(2.0f, c1[0], -x1.x); c1[1] = __builtin_fmaf(2.0f, c1[1], -x1.y); c1[2] = __builtin_fmaf(2.0f, c1[2], -x1.z); c1[3] = __builtin_fmaf(2.0f, c1[3], -x1.w);
        c2[0] = __builtin_fmaf(2.0f, c2[0], -x2.x); c2[1] = __builtin_fmaf(2.0f, c2[1], -x2.y); c2[2] = __builtin_fmaf(2.0f, c2[2], -x2.z); c2[3] = __builtin_fmaf(2.0f, c2[3], -x2.w);
        c3[0] = __builtin_fmaf(2.0f, c3[0], -x3.x); c3[1] = __builtin_fmaf(2.0f, c3[1], -x3.y); c3[2] = __builtin_fmaf(2.0f, c3[2], -x3.z); c3[3] = __builtin_fmaf(2.0f, c3[3], -x3.w);
    }
    // (a tree, not a chain: five independent v_max3 over fifteen values, then two, then one -- three dependent hops instead of eight)
    const float a0 = max3(c0[0], c0[1], c0[2]), a1 = max3(c0[3], c1[0], c1[1]), a2 = max3(c1[2], c1[3], c2[0]);
    const float a3 = max3(c2[1], c2[2], c2[3]), a4 = max3(c3[0], c3[1], c3[2]);
    const float b0 = max3(a0, a1, a2), b1 = max3(a3, a4, c3[3]);
    return max3(b0, b1, b1);
}
// the appends of a group in which some lane's maximum passed (rare: a group sees a survivor in ~7 % of the steady-state tiles):
// levels of wave ballots + scalar branches, as in qs_filter_tile
template <bool l2>
__device__ __forceinline__ void qs4_filter_slow(const f32x4 &c0, const f32x4 &c1, const f32x4 &c2, const f32x4 &c3, float th, unsigned &jb, float qn,
                                                int &ccnt, char *__restrict__ panel_bytes, unsigned my_off) {
    // (the row base is made opaque HERE: otherwise hipcc computes the sixteen row numbers of a group in front of the fast path,
    // every tile, for every group)
    asm volatile("" : "+v"(jb));
    const float m0 = max3(max3(c0[0], c0[1], c0[2]), c0[3], c0[3]), m1 = max3(max3(c1[0], c1[1], c1[2]), c1[3], c1[3]);
    const float m2 = max3(max3(c2[0], c2[1], c2[2]), c2[3], c2[3]), m3 = max3(max3(c3[0], c3[1], c3[2]), c3[3], c3[3]);
    const u64 bq[4] = {__ballot(m0 > th), __ballot(m1 > th), __ballot(m2 > th), __ballot(m3 > th)};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (bq[a]) {                                    // scalar branch
            const f32x4 &c = a == 0 ? c0 : a == 1 ? c1 : a == 2 ? c2 : c3;
            u64 be[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) be[i] = __ballot(c[i] > th);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (be[i]) {                            // scalar branch
                    if (c[i] > th) {                    // (rows >= n: -inf by the caller's masking MFMAs, last tile only)
                        const unsigned j = jb + 16 * a + i;
                        const float sc = l2 ? fminf(0.0f, c[i] - qn) : c[i];
                        *reinterpret_cast<u64 *>(panel_bytes + (my_off + 8u * append_slot4(ccnt))) = lemon_make_key(sc, j);
                        ++ccnt;
                    }
                }
            }
        }
    }
}

// entry e of a query's four quarter-lists read as ONE list of n0 + n1 + n2 + n3 keys (p1 = n0, p2 = n0 + n1, p3 = p2 + n2)
__device__ __forceinline__ u64 qs4_load(const u64 *__restrict__ list, int e, int p1, int p2, int p3, int ntot) {
    const int seg = (e >= p1) + (e >= p2) + (e >= p3);
    const int base = seg == 0 ? 0 : seg == 1 ? p1 : seg == 2 ? p2 : p3;
    return e < ntot ? list[seg * QCAP4 + (e - base)] : 0;
}

// light compaction of one query: bisection for (a lower bound of) the kk-th largest approximate score, survivors dealt
// back ROUND-ROBIN over the four quarters (entry i -> quarter i & 3, slot i >> 2).  Returns kept.
template <int NSL>
__device__ __forceinline__ int qs4_compact_light_ns(u64 *__restrict__ list, int p1, int p2, int p3, int ntot, int kk, float eps,
                                                    int lane, float *lo_out) {
    u64 v[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) v[i] = qs4_load(list, lane + 64 * i, p1, p2, p3, ntot);
    u32 o[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) o[i] = (u32)(v[i] >> 32);
    u32 t = 0;                                          // (see qs_compact_light_ns: any lower bound of tau keeps the band a proof)
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const u32 cand = t | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < NSL; ++i) c += __builtin_popcountll(__ballot(o[i] >= cand));
        if (c >= kk) {
            t = cand;
            if (c <= kk + 8) break;
        }
    }
    const float lo = bound_from_tau(lemon_ord2f(t), eps);
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int base = 0;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const bool keep = v[i] && lemon_key_score(v[i]) > lo;
        const u64 m = __ballot(keep);
        const int pos = base + __builtin_popcountll(m & below);
        if (keep) list[(pos & 3) * QCAP4 + (pos >> 2)] = v[i];        // (all loads precede these stores)
        base += __builtin_popcountll(m);
    }
    *lo_out = lo;
    return base;
}
__device__ __forceinline__ int qs4_compact_light(u64 *__restrict__ list, int n0, int n1, int n2, int n3, int kk, float eps, int lane,
                                                 float *lo_out) {
    const int p1 = n0, p2 = n0 + n1, p3 = p2 + n2, ntot = p3 + n3;
    const int ns = (ntot + 63) >> 6;                   // wave-uniform
    if (ns <= 2) return qs4_compact_light_ns<2>(list, p1, p2, p3, ntot, kk, eps, lane, lo_out);
    if (ns == 3) return qs4_compact_light_ns<3>(list, p1, p2, p3, ntot, kk, eps, lane, lo_out);
    if (ns == 4) return qs4_compact_light_ns<4>(list, p1, p2, p3, ntot, kk, eps, lane, lo_out);
    return qs4_compact_light_ns<8>(list, p1, p2, p3, ntot, kk, eps, lane, lo_out);
}

// exact compaction of one query whose `kept` keys sit round-robin in the quarters (right after a light compaction): exact
// fp32-chain scores, the exact top-kk dealt back round-robin.  Returns how many exist (<= kk).
__device__ __forceinline__ int qs4_compact_exact(const ScanParamsH &p, u64 *__restrict__ list, int kept, int64_t q, float qn, int lane,
                                                 u64 *__restrict__ sk, u64 *__restrict__ sb, u64 *kth_out) {
    const int kk = p.b.kk;
    const float *qrow = p.q + q * (int64_t)p.d;
    const bool l2 = p.b.metric == LEMON_METRIC_L2;
    u64 best = 0;
#pragma unroll 1
    for (int base = 0; base < kept; base += 64) {
        const int e = base + lane;
        const u64 old = e < kept ? list[(e & 3) * QCAP4 + (e >> 2)] : 0;
        u64 key = 0;
        if (old) {
            const u32 j = lemon_key_index(old);
            const float *xrow = p.x + (int64_t)j * p.d;
            const float s = exact_score(qrow, xrow, p.d, l2, qn, l2 ? p.b.xnorm[j] : 0.0f);
            key = (s == s) ? lemon_make_key(s, j) : 0;
        }
        const Ranked r = wave_rank_keys(best, key, 0, 0, 128, sk, lane);
        sb[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (best && r.r0 < kk) sb[r.r0] = best;
        if (key && r.r1 < kk) sb[r.r1] = key;
        __builtin_amdgcn_wave_barrier();
        best = sb[lane];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < kk) list[(lane & 3) * QCAP4 + (lane >> 2)] = best;
    *kth_out = __shfl(best, kk - 1);
    return __builtin_popcountll(__ballot(best != 0));
}

template <int KT, int PARK, bool l2>
__global__ __launch_bounds__(NT, 1) void k_scan_f16_qs4(ScanParamsH p) {
    // issue slots (MFMA index of a step behind which something else is issued): the five fragment reads of the next step behind
    // MFMAs RD0, RD0 + RDS, ...; the two DMA pieces of a step behind MFMAs DM0 and DM1
#ifndef LEMON_QS4_SLOTS
    constexpr int RD0 = 0, RDS = 1, DM0 = 8, DM1 = 12;
#else
    constexpr int RD0 = (LEMON_QS4_SLOTS) / 1000 % 10, RDS = (LEMON_QS4_SLOTS) / 100 % 10, DM0 = (LEMON_QS4_SLOTS) / 10 % 10 + 6, DM1 = (LEMON_QS4_SLOTS) % 10 + 6;
#endif
    constexpr int NS = 2 * KT;                 // k32 steps per tile
    constexpr int KT2 = KT / 2;                // stages per tile: 2 x 64-wide k-slices = 4 k32 steps each
    constexpr int NFR = 4 * NS - PARK;         // query fragments in registers: index f = b NS + s
    constexpr int NB = 4;                      // LDS stage ring
    constexpr int STG = 2 * RT2 * BK;          // floats per stage (16 KB)
    constexpr int PK = PARK ? PARK : 1;
    static_assert(NT == BQ2 && KT % 2 == 0 && PARK < NS && NFR >= 64, "fragment homes");
    __shared__ __attribute__((aligned(16))) float smem[NB * STG + (NT / 64) * PK * 256 + 2 * BQ2 + (NT / 64) * (512 + 128)];
    float *s_x = smem;                                   // [NB][2][64 * 32]
    float *s_q = smem + NB * STG;                        // [4 waves][PARK][64 lanes x 16 B] parked group-3 fragments
    float *s_qn = s_q + (NT / 64) * PK * 256;            // [256]
    float *s_eps = s_qn + BQ2;                           // [256]
    u64 *s_keys = reinterpret_cast<u64 *>(s_eps + BQ2);  // [4][256] rank-select scratch
    u64 *s_best = s_keys + (NT / 64) * 256;              // [4][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;

    const int panel = blockIdx.x / p.b.splits;
    const int64_t q0 = (int64_t)panel * BQ2;
    int t_begin = (blockIdx.x % p.b.splits) * p.b.tiles_per_split;
    int t_end = t_begin + p.b.tiles_per_split;
    if (t_end > p.b.n_tiles) t_end = p.b.n_tiles;
    if (p.b.splits == 1) { t_begin = p.chunk_t0; t_end = p.chunk_t1; }
    const int ntile = t_end - t_begin;
    const bool final_pass = (p.b.splits > 1) || p.last_chunk;
    const int dpad = p.dpad_h / 2;             // row pitch in 4-byte words

    {
        const float qn = p.b.qnorm[q0 + tid];
        s_eps[tid] = band_eps(p, qn, p.qres2[q0 + tid], l2);
        s_qn[tid] = qn;
    }

    // ---- lane-private candidate state per query group b: query = 64 wave + 16 b + (lane & 15), quarter g4 = lane >> 4 ----
    const int qrow0 = 64 * wave + l15;                   // group b: + 16 b
    u64 *cand_panel = p.b.cand + (int64_t)blockIdx.x * BQ2 * CAPH;
    char *panel_bytes = reinterpret_cast<char *>(cand_panel);
    const unsigned my_off0 = (unsigned)(qrow0 * CAPH + g4 * QCAP4) * 8u;      // group b: + b * 16 * CAPH * 8
    int ccnt0 = 0, ccnt1 = 0, ccnt2 = 0, ccnt3 = 0, clast0 = 0, clast1 = 0, clast2 = 0, clast3 = 0;
    float thk0, thk1, thk2, thk3, qn0 = 0.f, qn1 = 0.f, qn2 = 0.f, qn3 = 0.f;
    thk0 = (q0 + qrow0 < p.b.nq) ? -INFINITY : INFINITY;
    thk1 = (q0 + qrow0 + 16 < p.b.nq) ? -INFINITY : INFINITY;
    thk2 = (q0 + qrow0 + 32 < p.b.nq) ? -INFINITY : INFINITY;
    thk3 = (q0 + qrow0 + 48 < p.b.nq) ? -INFINITY : INFINITY;
    if (l2) { qn0 = p.b.qnorm[q0 + qrow0]; qn1 = p.b.qnorm[q0 + qrow0 + 16]; qn2 = p.b.qnorm[q0 + qrow0 + 32]; qn3 = p.b.qnorm[q0 + qrow0 + 48]; }
    if (p.b.splits == 1 && !p.first_chunk) {   // resume from the previous database chunk
        const float *st = p.state + 16 * ((int64_t)blockIdx.x * NT + tid);
        ccnt0 = __float_as_int(st[0]); clast0 = __float_as_int(st[1]); thk0 = st[2];
        ccnt1 = __float_as_int(st[4]); clast1 = __float_as_int(st[5]); thk1 = st[6];
        ccnt2 = __float_as_int(st[8]); clast2 = __float_as_int(st[9]); thk2 = st[10];
        ccnt3 = __float_as_int(st[12]); clast3 = __float_as_int(st[13]); thk3 = st[14];
    }
    auto th_of = [&](float tk, float qn) -> float {      // what the filter compares against (L2: proxy carries +|q|^2)
        if (!l2 || tk == -INFINITY || tk == INFINITY) return tk;
        return (tk + qn) - (fabsf(tk) + qn) * 2.4e-7f - 1e-37f;
    };
    float th0 = th_of(thk0, qn0), th1 = th_of(thk1, qn1), th2 = th_of(thk2, qn2), th3 = th_of(thk3, qn3);

    // ---- stationary operands: fragment (b, s) = the lane's query of group b, k = 32 s + 8 g4 .. + 7 ----
    bf16x8 qf[NFR];
    {
        const lp16 *src = p.qh + (q0 + qrow0) * (int64_t)p.dpad_h + 8 * g4;
#pragma unroll
        for (int s = NS - PARK; s < NS; ++s)   // (prologue: before any LDS-DMA is in flight)
            *reinterpret_cast<bf16x8 *>(s_q + ((wave * PK + (s - (NS - PARK))) * 64 + lane) * 4) =
                *reinterpret_cast<const bf16x8 *>(src + (int64_t)48 * p.dpad_h + 32 * s);
        __builtin_amdgcn_sched_barrier(0);
        // (loads in groups of eight with a scheduling fence in between: see k_scan_bf16_qs2)
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
            const int b = f / NS, s2 = f % NS;
            qf[f] = *reinterpret_cast<const bf16x8 *>(src + (int64_t)(16 * b) * p.dpad_h + 32 * s2);
            if ((f & 7) == 7 || f == NFR - 1) {
#pragma unroll
                for (int f2 = f & ~7; f2 <= f; ++f2) { if (f2 < 64) asm volatile("" : "+a"(qf[f2])); else asm volatile("" : "+v"(qf[f2])); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    f32x4 acc[4][4];                            // acc[a][b]: rows 16 a .. + 15 of the tile x query group b
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane (l15, g4) reads row 16 a + l15, 16-byte chunk 4 (s & 1) + g4 of slice (s >> 1) & 1: two base addresses (the swizzle
    // term repeats every 16 rows), a and the slice are immediate offsets, the ring slot is added per stage
    unsigned fa0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_x + swz(l15, g4));
    unsigned fa1 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_x + swz(l15, 4 + g4));
    const unsigned vq = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_q + (wave * PK * 64 + lane) * 4);

    const float *xbase = reinterpret_cast<const float *>(p.xh + (int64_t)t_begin * RT2 * p.dpad_h);
    const int total = ntile * KT2;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)s_x + (unsigned)wave * 2048u;
    const unsigned voff0 = (unsigned)(((16 * wave + (lane >> 3)) * dpad + 4 * ((lane & 7) ^ ((lane >> 4) & 7))) * 4);
    const unsigned voff1 = (unsigned)(((16 * wave + 8 + (lane >> 3)) * dpad + 4 * ((lane & 7) ^ ((4 + (lane >> 4)) & 7))) * 4);
    // piece pc (0..3) of the stage kt2_ of the tile at tile_base into ring slot slot_: slice pc >> 1, row half pc & 1
#define Q4_PIECE(tile_base, kt2_, slot_, pc)                                                                               \
    qs4_dma_piece((tile_base) + (2 * (kt2_) + ((pc) >> 1)) * BK, lds0 + (unsigned)(((slot_) * STG + ((pc) >> 1) * RT2 * BK) * 4) + 1024u * ((pc) & 1), \
                  ((pc) & 1) ? voff1 : voff0)
#pragma unroll
    for (int s0 = 0; s0 < NB - 1; ++s0)
        if (s0 < total) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) Q4_PIECE(xbase + (int64_t)(s0 / KT2) * RT2 * dpad, s0 % KT2, s0, pc);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (prologue only)
    __syncthreads();

    bf16x8 fA[2][4], fP[2];                     // two fragment sets: database fragments a = 0..3 (+ the parked query fragment)
#pragma unroll
    for (int u = 0; u < 2; ++u) { fP[u] = bf16x8{}; for (int a = 0; a < 4; ++a) fA[u][a] = bf16x8{}; }
    unsigned va0 = fa0, va1 = fa1;              // fragment addresses inside the ring slot being read (slot 0 first)
    const bool filter_on = !(p.ablate & 1);
    if (p.ablate & 4) th0 = th1 = th2 = th3 = INFINITY;      // (diagnostic, results invalid: nothing passes the filter, so nothing re-arms it)
    const int lim4 = (p.b.stale + 3) >> 2;                   // a lane's share of `stale` new candidates per query

    // one database fragment of k32 step S (of the tile) into set U: address base VA0 / VA1 = the stage's slot
#define Q4_LOADA(U, S, a_)                                                                                                 \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fA[U][a_]) : "v"(((S) & 1) ? va1 : va0), "n"((((S) >> 1) & 1) * RT2 * 128 + 2048 * (a_)) : "memory")
#define Q4_LOADP(U, S)                                                                                                     \
    do { if (PARK && (S) >= NS - PARK) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fP[U]) : "v"(vq), "n"(((S) >= NS - PARK ? (S) - (NS - PARK) : 0) * 1024) : "memory"); } while (0)
#define Q4_WAIT(U) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fA[U][0]), "+v"(fA[U][1]), "+v"(fA[U][2]), "+v"(fA[U][3]), "+v"(fP[U]))
    // MFMA i of step S: query group b = i >> 2 against database fragment a = i & 3
#define Q4_MF(U, S, i_)                                                                                                    \
    do {                                                                                                                   \
        constexpr int b_ = (i_) >> 2, a_ = (i_) & 3, f_ = b_ * NS + (S);                                                   \
        if (f_ >= NFR)      mfma16<false, (S) == 0>(acc[a_][b_], fA[U][a_], fP[U]);                                        \
        else if (f_ < 64)   mfma16<true, (S) == 0>(acc[a_][b_], fA[U][a_], qf[f_ < NFR ? f_ : 0]);                         \
        else                mfma16<false, (S) == 0>(acc[a_][b_], fA[U][a_], qf[f_ < NFR ? f_ : 0]);                        \
    } while (0)

    // ---- maintenance: light compaction of the queries whose lists need it (all four groups; b, c wave-uniform) ----
#define Q4_SEL(b_, x0, x1, x2, x3) ((b_) == 0 ? (x0) : (b_) == 1 ? (x1) : (b_) == 2 ? (x2) : (x3))
    // `todo`: bit 16 b + c = query c of group b asked for it (Q4_SLOW, right behind the appends that brought it there).  The
    // triggers are PER LANE -- my quarter grew by a quarter of `stale` since the last compaction, or is within one tile of its
    // capacity, or holds a first tile's worth while the query has no bound yet -- so that a tile with appends costs no cross-lane
    // traffic at all (round-5 measurement: with the four-lane sums of QS2's rule made after every tile that appended anything, the
    // appends + compactions + the barrier skew they cause took 15 % of the 1 M x 768 scan, against 10 % in QS2; any compaction
    // cadence gives the same bits, the final selection is exact)
    auto maintain = [&](u64 todo) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's appends are visible
        do {
            const int r = __ffsll((long long)todo) - 1;     // = 16 b + c: the query's row inside the wave's 64
            todo &= todo - 1;
            const int b = r >> 4, c = r & 15;
            const int row = 64 * wave + r;
            u64 *list = cand_panel + (int64_t)row * CAPH;
            const int cc = Q4_SEL(b, ccnt0, ccnt1, ccnt2, ccnt3);
            const int n0 = __builtin_amdgcn_readlane(cc, c), n1 = __builtin_amdgcn_readlane(cc, c + 16);
            const int n2 = __builtin_amdgcn_readlane(cc, c + 32), n3 = __builtin_amdgcn_readlane(cc, c + 48);
            if (n0 + n1 + n2 + n3 < p.b.kk) continue;   // (a lane trigger before the query holds kk keys: nothing to select yet)
            float lo;
            int kept = qs4_compact_light(list, n0, n1, n2, n3, p.b.kk, s_eps[row], lane, &lo);
            if (kept > CAPH - 64) {            // the band itself leaves no room for a tile's appends: settle it exactly
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                u64 kth;
                kept = qs4_compact_exact(p, list, kept, q0 + row, s_qn[row], lane, s_keys + wave * 256, s_best + wave * 64, &kth);
                if (kept == p.b.kk) lo = bound_from_tau(lemon_key_score(kth), s_eps[row]);
            }
            const int mine = (kept + 3 - g4) >> 2;        // round-robin deal: quarter g holds entries g, g + 4, ...
            const bool me = l15 == c;
            if (me && b == 0) { ccnt0 = mine; clast0 = mine; thk0 = lo; th0 = th_of(lo, qn0); }
            if (me && b == 1) { ccnt1 = mine; clast1 = mine; thk1 = lo; th1 = th_of(lo, qn1); }
            if (me && b == 2) { ccnt2 = mine; clast2 = mine; thk2 = lo; th2 = th_of(lo, qn2); }
            if (me && b == 3) { ccnt3 = mine; clast3 = mine; thk3 = lo; th3 = th_of(lo, qn3); }
        } while (todo);
    };
    // the filter of the finished tile whose first row is jt_, per query group b_: Q4_GMAX = the group's pass mask (a wave ballot of
    // "my maximum beats my threshold"), Q4_SLOW = its appends when the mask is not empty (wave-uniform `any` |= appended).  In the
    // loop a group's mask is made one group AHEAD of its branch (compare -> scalar branch is a dependent hop the wave would wait on)
#define Q4_ACC(b_) acc[0][b_], acc[1][b_], acc[2][b_], acc[3][b_]
#define Q4_TH(b_) ((b_) == 0 ? th0 : (b_) == 1 ? th1 : (b_) == 2 ? th2 : th3)
#define Q4_GMAX(b_, jt_) (__ballot(qs4_group_max<l2>(Q4_ACC(b_), (jt_) + 4u * (unsigned)g4, p.b.xnorm) > Q4_TH(b_)))
    // (a lane whose query does not exist -- the panel's padding -- carries the bound +inf: it never appends, its counts stay 0)
#define Q4_NEED(b_, cc_, cl_, tk_) ((cc_) - (cl_) >= lim4 || (cc_) > QCAP4 - 16 || ((tk_) == -INFINITY && (cc_) >= 16))
#define Q4_SLOW(b_, jt_)                                                                                                   \
    do {                                                                                                                   \
        unsigned jb_ = (jt_) + 4u * (unsigned)g4;                                                                          \
        u64 nd_ = 0;                                                                                                       \
        if ((b_) == 0) { qs4_filter_slow<l2>(Q4_ACC(0), Q4_TH(0), jb_, qn0, ccnt0, panel_bytes, my_off0); nd_ = __ballot(Q4_NEED(0, ccnt0, clast0, thk0)); } \
        if ((b_) == 1) { qs4_filter_slow<l2>(Q4_ACC(1), Q4_TH(1), jb_, qn1, ccnt1, panel_bytes, my_off0 + 16u * CAPH * 8u); nd_ = __ballot(Q4_NEED(1, ccnt1, clast1, thk1)); } \
        if ((b_) == 2) { qs4_filter_slow<l2>(Q4_ACC(2), Q4_TH(2), jb_, qn2, ccnt2, panel_bytes, my_off0 + 32u * CAPH * 8u); nd_ = __ballot(Q4_NEED(2, ccnt2, clast2, thk2)); } \
        if ((b_) == 3) { qs4_filter_slow<l2>(Q4_ACC(3), Q4_TH(3), jb_, qn3, ccnt3, panel_bytes, my_off0 + 48u * CAPH * 8u); nd_ = __ballot(Q4_NEED(3, ccnt3, clast3, thk3)); } \
        todo |= ((nd_ | (nd_ >> 16) | (nd_ >> 32) | (nd_ >> 48)) & 0xffffull) << (16 * (b_));                              \
    } while (0)

    // k32 step S (compile-time) of the tile on fragment set S & 1.  In the gaps between its MFMAs: the reads of step S + 1
    // (the next stage's slot once S + 1 starts a stage), in steps 2 and 3 of a stage the DMA pieces of stage t + 3, and in
    // step 0 of a tile the filter of the previous tile (group b in front of group b's first MFMA).
    // ---- the filter's fast path, one instruction per MFMA gap (inner-product metric) ----
    // Phase stamps of the first QS4 build (tools/r5_build_phases.sh, 1 M x 768): step 0 of a tile -- 16 MFMAs = 256 cycles -- took 770
    // with the four groups' maximum trees in front of their MFMAs: everything the wave issues there is on its critical path (one wave
    // per SIMD), 8 cycles per instruction with the dependent tree levels.  So the maximum of group b is made as a CHAIN of eight
    // v_max3_f32, one per gap, from the moment the group's last MFMAs of the tile's LAST step have drained (gap G0 = 5 + 4 b behind
    // MFMA 0 of that step; the chain reads the rows of fragment a = 3, written last, from its sixth link on) into the first step of
    // the next tile, where the ballot and the branch of group b stand in front of the group's first MFMA (gaps 16 + 4 b).
#define Q4_V(b_, e_) acc[(e_) >> 2][b_][(e_) & 3]
#define Q4_TREE1(b_, k_)                                                                                                   \
    do {                                                                                                                   \
        float &m_ = (b_) == 0 ? mt0 : (b_) == 1 ? mt1 : (b_) == 2 ? mt2 : mt3;                                             \
        if ((k_) == 0)      asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m_) : "v"(Q4_V(b_, 0)), "v"(Q4_V(b_, 1)), "v"(Q4_V(b_, 2))); \
        else if ((k_) < 7)  asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m_) : "v"(Q4_V(b_, 2 * (k_) + 1)), "v"(Q4_V(b_, (2 * (k_) + 2) & 15))); \
        else if ((k_) == 7) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(m_) : "v"(Q4_V(b_, 15)));                      \
        else {                                                                                                             \
            const u64 bl_ = __ballot(m_ > Q4_TH(b_));                                                                      \
            if ((b_) == 0) pm0 = bl_; else if ((b_) == 1) pm1 = bl_; else if ((b_) == 2) pm2 = bl_; else pm3 = bl_;        \
        }                                                                                                                  \
    } while (0)
    // gap g_ (0..15: behind MFMA g_ of the tile's last step; 16..31: behind MFMA g_ - 16 of the next tile's first step)
#define Q4_TREES(g_)                                                                                                       \
    do {                                                                                                                   \
        if ((g_) >= 5 && (g_) <= 13)  Q4_TREE1(0, (g_) - 5);                                                               \
        if ((g_) >= 9 && (g_) <= 17)  Q4_TREE1(1, (g_) - 9);                                                               \
        if ((g_) >= 13 && (g_) <= 21) Q4_TREE1(2, (g_) - 13);                                                              \
        if ((g_) >= 17 && (g_) <= 25) Q4_TREE1(3, (g_) - 17);                                                              \
    } while (0)

    // (written out slot by slot, no loop for hipcc to unroll: where it declined to -- the L2 instantiation at pitch 512 once --
    // the fragment registers of in-flight asm reads were copied around a 16-way switch; tests/test_build_guard.py)
#define Q4_SLOT(S, i_)                                                                                                     \
    do {                                                                                                                   \
        if ((S) == 0 && ((i_) & 3) == 0 && do_filter) {                                                                    \
            if (l2) {   /* (the L2 filter -- |x|^2 loads, proxy transform -- stays in front of its group's first MFMA) */       \
                if ((i_) == 0)  { pm0 = Q4_GMAX(0, jprev); pm1 = Q4_GMAX(1, jprev); if (pm0) Q4_SLOW(0, jprev); }          \
                if ((i_) == 4)  { pm2 = Q4_GMAX(2, jprev); if (pm1) Q4_SLOW(1, jprev); }                                   \
                if ((i_) == 8)  { pm3 = Q4_GMAX(3, jprev); if (pm2) Q4_SLOW(2, jprev); }                                   \
                if ((i_) == 12) { if (pm3) Q4_SLOW(3, jprev); }                                                            \
            } else {    /* (the group maxima were made between the MFMAs of the previous step and of this one: Q4_TREES) */    \
                if ((i_) == 0  && pm0) Q4_SLOW(0, jprev);                                                                  \
                if ((i_) == 4  && pm1) Q4_SLOW(1, jprev);                                                                  \
                if ((i_) == 8  && pm2) Q4_SLOW(2, jprev);                                                                  \
                if ((i_) == 12 && pm3) Q4_SLOW(3, jprev);                                                                  \
            }                                                                                                              \
        }                                                                                                                  \
        Q4_MF((S) & 1, S, i_);                                                                                             \
        if (!l2 && (S) == NS - 1) Q4_TREES(i_);                                                                            \
        if (!l2 && (S) == 0) Q4_TREES(16 + (i_));                                                                          \
        if ((i_) == RD0) Q4_LOADA(((S) & 1) ^ 1, ((S) + 1) % NS, 0);                                                       \
        if ((i_) == RD0 + RDS) Q4_LOADA(((S) & 1) ^ 1, ((S) + 1) % NS, 1);                                                 \
        if ((i_) == RD0 + 2 * RDS) Q4_LOADA(((S) & 1) ^ 1, ((S) + 1) % NS, 2);                                             \
        if ((i_) == RD0 + 3 * RDS) Q4_LOADA(((S) & 1) ^ 1, ((S) + 1) % NS, 3);                                             \
        if ((i_) == RD0 + 4 * RDS) Q4_LOADP(((S) & 1) ^ 1, ((S) + 1) % NS);                                                \
        if (((S) & 3) >= 2 && ((i_) == DM0 || (i_) == DM1)) {                                                              \
            /* piece pc of stage t + 3: (add,) M0, nop, load -- source and destination base were made per stage */          \
            constexpr int pc_ = 2 * (((S) & 3) - 2) + ((i_) == DM1);                                                       \
            qs4_dma_piece_off<(pc_ >> 1) * BK * 4>(ssrc_, pc_ == 0 ? sl0_ : pc_ == 1 ? sl1_ : pc_ == 2 ? sl2_ : sl3_, (pc_ & 1) ? voff1 : voff0); \
        }                                                                                                                  \
    } while (0)
#define Q4_STEP(S)                                                                                                         \
    do {                                                                                                                   \
        Q4_WAIT((S) & 1);                                                                                                  \
        if (((S) & 3) == 3) {   /* the next step opens a stage: its slot */                                                \
            const unsigned sbn_ = (unsigned)(((t + 1) & (NB - 1)) * STG * 4);                                              \
            va0 = fa0 + sbn_; va1 = fa1 + sbn_;                                                                            \
        }                                                                                                                  \
        Q4_SLOT(S, 0); Q4_SLOT(S, 1); Q4_SLOT(S, 2); Q4_SLOT(S, 3); Q4_SLOT(S, 4); Q4_SLOT(S, 5); Q4_SLOT(S, 6); Q4_SLOT(S, 7);    \
        Q4_SLOT(S, 8); Q4_SLOT(S, 9); Q4_SLOT(S, 10); Q4_SLOT(S, 11); Q4_SLOT(S, 12); Q4_SLOT(S, 13); Q4_SLOT(S, 14); Q4_SLOT(S, 15); \
        if (((S) & 3) == 1) {                                                                                              \
            /* stage t + 1 has landed (this wave's pieces: at most the four of stage t + 2 may still be in flight; younger */ \
            /* appends only make the wait longer), then the rendezvous: everybody's pieces, and everybody is done with stage t - 1 */ \
            Q4_PH_BEGIN();                                                                                                 \
            if (t + 2 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                            \
            else               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
            __builtin_amdgcn_s_barrier();                                                                                  \
            Q4_PH_END(3);                                                                                                  \
        }                                                                                                                  \
    } while (0)

    // the first step's fragments (slot 0)
    if (ntile > 0) {
        Q4_LOADA(0, 0, 0); Q4_LOADA(0, 0, 1); Q4_LOADA(0, 0, 2); Q4_LOADA(0, 0, 3);
    }
    // (stages written out with literal indices: fragment homes and offsets are template / immediate operands)
#ifdef LEMON_QS4_PHASES
    // diagnostic build (tools/r5_build_phases.sh; results unchanged, timing +~10 %): shader-cycle sums per wave 0 of every workgroup --
    // [0] whole loop, [1] step 0 of every tile (the 16 init MFMAs + the previous tile's filter), [2] maintain(), [3] the wait +
    // barrier of every stage, [4] tiles, [5] tiles whose filter took a slow path, [6] tiles with a compaction
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_ = 0, pl_ = __builtin_amdgcn_s_memtime();
#define Q4_PH_BEGIN() do { pt_ = __builtin_amdgcn_s_memtime(); } while (0)
#define Q4_PH_END(i_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_[i_] += n_ - pt_; pt_ = n_; } while (0)
#else
#define Q4_PH_BEGIN() do { } while (0)
#define Q4_PH_END(i_) do { } while (0)
#endif
#define Q4_STAGE(K)                                                                                                        \
    do {                                                                                                                   \
        const int t = jl * KT2 + (K);                                                                                      \
        /* stage t + 3's pieces (issued in steps 2, 3): ALWAYS issued -- behind the end of the launch they re-fetch the launch's  \
           first stage into a ring slot nobody reads any more -- so that no branch stands between the MFMAs; source pointer and    \
           the four LDS destinations live in SGPRs from here (made once per stage, not per piece) */                             \
        const float *ssrc_ = (t + 3 < total) ? xt + (int64_t)(((K) + 3) / KT2) * RT2 * dpad + 2 * (((K) + 3) % KT2) * BK : xbase; \
        unsigned sl0_ = lds0 + (unsigned)(((t + 3) & (NB - 1)) * STG * 4);                                                 \
        asm volatile("" : "+s"(ssrc_), "+s"(sl0_));     /* (three SGPRs: the kernel has none to spare for all four destinations) */ \
        /* (pieces 2, 3 carry the k-slice's 128 source bytes in the instruction's offset field, and LDS-DMA adds that field to   \
           the LDS address as well: their M0 is 128 short) */                                                                  \
        const unsigned sl1_ = sl0_ + 1024u, sl2_ = sl0_ + (unsigned)(RT2 * BK * 4 - BK * 4), sl3_ = sl0_ + (unsigned)(RT2 * BK * 4 - BK * 4) + 1024u; \
        if ((K) == 0) Q4_PH_BEGIN();                                                                                       \
        Q4_STEP(4 * (K) + 0);                                                                                              \
        if ((K) == 0) Q4_PH_END(1);                                                                                        \
        if ((K) == 0 && todo) { maintain(todo); Q4_PH_END(2); }                                                            \
        Q4_STEP(4 * (K) + 1);                                                                                              \
        Q4_STEP(4 * (K) + 2);                                                                                              \
        Q4_STEP(4 * (K) + 3);                                                                                              \
    } while (0)
    u64 pm0 = 0, pm1 = 0, pm2 = 0, pm3 = 0;     // a group's pass mask (ballot of "my maximum beats my bound"), made by Q4_TREES / Q4_GMAX
    float mt0 = 0.f, mt1 = 0.f, mt2 = 0.f, mt3 = 0.f;
    for (int jl = 0; jl < ntile; ++jl) {
        const float *xt = xbase + (int64_t)jl * RT2 * dpad;
        const bool do_filter = filter_on && jl > 0;
        const unsigned jprev = (unsigned)(t_begin + jl - 1) * RT2;
        u64 todo = 0;
        Q4_STAGE(0); Q4_STAGE(1); Q4_STAGE(2); Q4_STAGE(3);
        if constexpr (KT2 > 4) { Q4_STAGE(4); Q4_STAGE(5); }
#ifdef LEMON_QS4_PHASES
        ph_[4] += 1; ph_[5] += (pm0 | pm1 | pm2 | pm3) != 0; ph_[6] += todo != 0;
#endif
        static_assert(KT2 == 4 || KT2 == 6, "stages per tile written out for d = 512 and d = 768");
    }
#undef Q4_STAGE
#undef Q4_PH_BEGIN
#undef Q4_PH_END
    Q4_WAIT(0);                                 // (the reads the last step issued for a tile that does not exist: retired, unused)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (and the pieces issued behind the end of the launch)
#ifdef LEMON_QS4_PHASES
    ph_[0] = __builtin_amdgcn_s_memtime() - pl_;
    if (p.phase_dbg && tid == 0) { for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&p.phase_dbg[i_], ph_[i_]); }
#endif
#undef Q4_STEP
#undef Q4_SLOT
#undef Q4_TREES
#undef Q4_TREE1
#undef Q4_V
#undef Q4_MF
#undef Q4_WAIT
#undef Q4_LOADP
#undef Q4_LOADA
#undef Q4_PIECE
    // ---- the last tile of the launch: its filter has no next tile to ride on ----
    if (ntile > 0 && filter_on) {
        const unsigned jt = (unsigned)(t_end - 1) * RT2;
        if ((unsigned)t_end * RT2 > (unsigned)p.b.n) {
            // last tile of the database: its padding rows must never pass.  One more MFMA per accumulator tile: A' = -inf in
            // k-slot 0 of the padding rows (0 elsewhere), B' = 1 in k-slot 0 -> padding rows' scores become -inf, valid rows' + 0
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 ones = {}, mk0 = {}, mk1 = {}, mk2 = {}, mk3 = {};
            ones[0] = g4 == 0 ? (unsigned short)0x3c00u : (unsigned short)0;
            mk0[0] = (g4 == 0 && jt + (unsigned)l15 >= (unsigned)p.b.n) ? (unsigned short)0xfc00u : (unsigned short)0;
            mk1[0] = (g4 == 0 && jt + 16u + (unsigned)l15 >= (unsigned)p.b.n) ? (unsigned short)0xfc00u : (unsigned short)0;
            mk2[0] = (g4 == 0 && jt + 32u + (unsigned)l15 >= (unsigned)p.b.n) ? (unsigned short)0xfc00u : (unsigned short)0;
            mk3[0] = (g4 == 0 && jt + 48u + (unsigned)l15 >= (unsigned)p.b.n) ? (unsigned short)0xfc00u : (unsigned short)0;
            bf16x8 b1 = __builtin_bit_cast(bf16x8, ones), a0 = __builtin_bit_cast(bf16x8, mk0), a1 = __builtin_bit_cast(bf16x8, mk1);
            bf16x8 a2 = __builtin_bit_cast(bf16x8, mk2), a3 = __builtin_bit_cast(bf16x8, mk3);
            // (VALU -> asm MFMA: hipcc pads nothing in front of an asm statement)
            asm volatile("s_nop 4" : "+v"(b1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                mfma16<false, false>(acc[0][b], a0, b1); mfma16<false, false>(acc[1][b], a1, b1);
                mfma16<false, false>(acc[2][b], a2, b1); mfma16<false, false>(acc[3][b], a3, b1);
            }
        }
        // MFMA results are read by VALU next: wait out the matrix pipe (hipcc pads nothing around asm)
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]));
        asm volatile("" : "+v"(acc[0][1]), "+v"(acc[1][1]), "+v"(acc[2][1]), "+v"(acc[3][1]));
        asm volatile("" : "+v"(acc[0][2]), "+v"(acc[1][2]), "+v"(acc[2][2]), "+v"(acc[3][2]));
        asm volatile("" : "+v"(acc[0][3]), "+v"(acc[1][3]), "+v"(acc[2][3]), "+v"(acc[3][3]));
        u64 todo = 0;
        if (Q4_GMAX(0, jt)) Q4_SLOW(0, jt);
        if (Q4_GMAX(1, jt)) Q4_SLOW(1, jt);
        if (Q4_GMAX(2, jt)) Q4_SLOW(2, jt);
        if (Q4_GMAX(3, jt)) Q4_SLOW(3, jt);
        if (todo) maintain(todo);
    }
#undef Q4_SLOW
#undef Q4_NEED
#undef Q4_GMAX
#undef Q4_TH
#undef Q4_ACC
#undef Q4_SEL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!final_pass) {      // park the lane-private state for the next database chunk
        float *st = p.state + 16 * ((int64_t)blockIdx.x * NT + tid);
        st[0] = __int_as_float(ccnt0); st[1] = __int_as_float(clast0); st[2] = thk0;
        st[4] = __int_as_float(ccnt1); st[5] = __int_as_float(clast1); st[6] = thk1;
        st[8] = __int_as_float(ccnt2); st[9] = __int_as_float(clast2); st[10] = thk2;
        st[12] = __int_as_float(ccnt3); st[13] = __int_as_float(clast3); st[14] = thk3;
        return;
    }
    // ---- end of the scan: counts for k_bf16_final (exact re-scoring + exact top-k, one wave per query) ----
    int *cn = p.cnt + 4 * ((int64_t)blockIdx.x * BQ2 + qrow0) + g4;
    cn[0] = ccnt0; cn[4 * 16] = ccnt1; cn[4 * 32] = ccnt2; cn[4 * 48] = ccnt3;
}

// ======================================================================================
// Final pass of the Q-stationary scans: exact fp32-chain re-scoring of every surviving candidate + exact top-k.
//
// Inside the scan kernel this ran with ONE wave per SIMD, one lane per candidate row, every lane walking its own 3 KB row
// 16 B at a time: 64 cache lines per load instruction, four waves of loads in flight per CU.  Measured at 1 M x 768 (kernel
// trace, 23 chunk launches per 524 288 queries): steady-state launch 28 ms, the LAST launch 87-93 ms -- the final pass was
// 8.6 % of the whole scan.  Here one wave owns one query, many waves share a CU, candidate rows are fetched a full 128-B line
// per 8 lanes and transposed through LDS (the K4 gather's scheme), the query row sits in LDS once and is read by broadcast.
// Same fmaf chain in ascending k as exact_score(): bit-identical keys.
// ======================================================================================
constexpr int FIN_PITCH = 36;      // floats per transposed row (conflict-free 128-bit reads)

struct FinalParams {
    ScanParams b;              // D / I / part / nq / kk / metric / splits / nq_pad / xnorm / qnorm / cand
    const float *q, *x;        // originals, row-major [nq, d], [n, d]
    const int *cnt;            // [lists][segs]
    int d, rows_per_wg;        // queries per scan workgroup (128 / 256)
    int segs;                  // lane lists per query: 2 halves of 256 (QS, QS2) or 4 quarters of 128 (QS4)
    int64_t n_lists;
    const float *qres2;        // [nq_pad] measured ||q - qh||^2 (band_eps)
    const unsigned *xstat;     // database maxima (band_eps)
    int prefilter;             // 1: a last light compaction on the approximate keys before any row is fetched
};

// dot(q, x_row) by the chain contract for 64 rows at once: lane l owns x_row (its candidate); rows are fetched one 128-B line
// at a time -- lane (r8, c) fetches chunk c of rows r8, r8 + 8, ... -- transposed through `lx` [64][FIN_PITCH], and the next
// line is in flight while the current one is consumed against q_lds (broadcast reads).  d % 4 == 0, 16-byte aligned rows.
__device__ __forceinline__ float chain_dot_wave_q(const float *__restrict__ q_lds, const float *x_row, int d,
                                                  float *__restrict__ lx, int lane, int nvalid) {
    const int r8 = lane >> 3, c = lane & 7;       // (nvalid: lanes >= nvalid own no candidate -- their rows are not fetched)
#define FIN_PTR(i) const float *px##i = reinterpret_cast<const float *>(__shfl((unsigned long long)(uintptr_t)x_row, r8 + 8 * i)) + 4 * c;
    FIN_PTR(0) FIN_PTR(1) FIN_PTR(2) FIN_PTR(3) FIN_PTR(4) FIN_PTR(5) FIN_PTR(6) FIN_PTR(7)
#undef FIN_PTR
    // two register sets: the rows' next TWO 128-B lines are in flight while the current one is consumed (with one set the
    // gather ran at 3.5 TB/s: 8 KB in flight per wave)
    float4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#define FIN_FETCH(S, koff)                                                                          \
    do {                                                                                            \
        S##0 = S##1 = S##2 = S##3 = S##4 = S##5 = S##6 = S##7 = zero;                               \
        if ((koff) + 4 * c < d) {                                                                   \
            if (r8 < nvalid)      S##0 = *reinterpret_cast<const float4 *>(px0 + (koff));          \
            if (r8 + 8 < nvalid)  S##1 = *reinterpret_cast<const float4 *>(px1 + (koff));          \
            if (r8 + 16 < nvalid) S##2 = *reinterpret_cast<const float4 *>(px2 + (koff));          \
            if (r8 + 24 < nvalid) S##3 = *reinterpret_cast<const float4 *>(px3 + (koff));          \
            if (r8 + 32 < nvalid) S##4 = *reinterpret_cast<const float4 *>(px4 + (koff));          \
            if (r8 + 40 < nvalid) S##5 = *reinterpret_cast<const float4 *>(px5 + (koff));          \
            if (r8 + 48 < nvalid) S##6 = *reinterpret_cast<const float4 *>(px6 + (koff));          \
            if (r8 + 56 < nvalid) S##7 = *reinterpret_cast<const float4 *>(px7 + (koff));          \
        }                                                                                           \
    } while (0)
#define FIN_ST1(i, v) *reinterpret_cast<float4 *>(&lx[(r8 + 8 * i) * FIN_PITCH + 4 * c]) = v;
#define FIN_STORE(S) FIN_ST1(0, S##0) FIN_ST1(1, S##1) FIN_ST1(2, S##2) FIN_ST1(3, S##3) FIN_ST1(4, S##4) FIN_ST1(5, S##5) FIN_ST1(6, S##6) FIN_ST1(7, S##7)
#define FIN_CONSUME(k0_)                                                                            \
    do {                                                                                            \
        __builtin_amdgcn_wave_barrier();                                                            \
        const int lim = d - (k0_) < 32 ? d - (k0_) : 32;       /* wave-uniform; multiple of 4 */     \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                             \
            if (4 * u < lim) {                                                                      \
                const float4 xv = *reinterpret_cast<const float4 *>(&lx[lane * FIN_PITCH + 4 * u]); \
                const float4 qv = *reinterpret_cast<const float4 *>(&q_lds[(k0_) + 4 * u]);         \
                acc = __builtin_fmaf(qv.x, xv.x, acc);                                              \
                acc = __builtin_fmaf(qv.y, xv.y, acc);                                              \
                acc = __builtin_fmaf(qv.z, xv.z, acc);                                              \
                acc = __builtin_fmaf(qv.w, xv.w, acc);                                              \
            }                                                                                       \
        }                                                                                           \
        __builtin_amdgcn_wave_barrier();                                                            \
    } while (0)
    FIN_FETCH(a, 0);
    FIN_FETCH(b, 32);
    float acc = 0.0f;
    for (int k0 = 0; k0 < d; k0 += 64) {
        FIN_STORE(a)
        FIN_FETCH(a, k0 + 64);                           // (past the end: zeros, no loads)
        FIN_CONSUME(k0);
        if (k0 + 32 < d) {                               // wave-uniform
            FIN_STORE(b)
            FIN_FETCH(b, k0 + 96);
            FIN_CONSUME(k0 + 32);
        }
    }
#undef FIN_CONSUME
#undef FIN_STORE
#undef FIN_ST1
#undef FIN_FETCH
    return acc;
}

// A query's list as the scan leaves it holds the survivors of its LAST light compaction plus everything appended since (up to
// `stale` keys, ~100 entries in all at k = 51), but only the keys within the band of the FINAL k-th best approximate score can
// be in the exact top-k (~60): one more light compaction -- the scan's own rule, no database access -- before the gather cuts the
// rows k_bf16_final fetches by a third.  Survivors go to `sk` (256 keys of LDS); returns their number, or -1 when they do not
// fit (band-crowded data: the caller walks the list itself).
template <int NSL>
__device__ __forceinline__ int final_prefilter_ns(const u64 *__restrict__ list, int seg_cap, int p1, int p2, int p3, int c, int kk, float eps,
                                                  int lane, u64 *__restrict__ sk) {
    u64 v[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int e = lane + 64 * i;
        const int seg = (e >= p1) + (e >= p2) + (e >= p3);
        const int sbase = seg == 0 ? 0 : seg == 1 ? p1 : seg == 2 ? p2 : p3;
        v[i] = e < c ? list[seg * seg_cap + (e - sbase)] : 0;
    }
    u32 t = 0;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const u32 cand = t | (1u << bit);
        int n = 0;
#pragma unroll
        for (int i = 0; i < NSL; ++i) n += __builtin_popcountll(__ballot((u32)(v[i] >> 32) >= cand));
        if (n >= kk) {
            t = cand;
            if (n <= kk + 2) break;
        }
    }
    const float lo = bound_from_tau(lemon_ord2f(t), eps);
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int kept = 0;
#pragma unroll
    for (int i = 0; i < NSL; ++i) kept += __builtin_popcountll(__ballot(v[i] && lemon_key_score(v[i]) > lo));
    if (kept > 256) return -1;
    int base = 0;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const bool keep = v[i] && lemon_key_score(v[i]) > lo;
        const u64 m = __ballot(keep);
        if (keep) sk[base + __builtin_popcountll(m & below)] = v[i];
        base += __builtin_popcountll(m);
    }
    return kept;
}

// one wave per candidate list (= per query, or per (query, database split)); STAGED needs d % 4 == 0
template <bool l2, bool STAGED>
__global__ __launch_bounds__(256) void k_bf16_final(FinalParams p) {
    constexpr int DQ = 1024;                              // query row capacity in LDS (register-resident pitches: d <= 768)
    __shared__ __attribute__((aligned(16))) float s_q[4][STAGED ? DQ : 4];
    __shared__ __attribute__((aligned(16))) float s_x[4][STAGED ? 64 * FIN_PITCH : 4];
    __shared__ __attribute__((aligned(16))) u64 s_keys[4][256];
    __shared__ __attribute__((aligned(16))) u64 s_best[4][64];
    __shared__ __attribute__((aligned(16))) u64 s_surv[4][256];   // the list after the last light compaction (final_prefilter_ns)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t L = (int64_t)blockIdx.x * 4 + wave;     // list = (scan workgroup, row inside it)
    if (L >= p.n_lists) return;
    const int64_t wg = L / p.rows_per_wg;
    const int row = (int)(L - wg * p.rows_per_wg);
    const int64_t panel = wg / p.b.splits;
    const int split = (int)(wg - panel * p.b.splits);
    const int64_t q = panel * p.rows_per_wg + row;
    if (q >= p.b.nq) return;
    const u64 *list = p.b.cand + L * CAPH;
    // the query's lane lists read as one: segment s holds cnt[s] keys at list[s * CAPH / segs ...]; p1..p3 = prefix counts
    const int segs = p.segs, seg_cap = CAPH / segs;
    const int p1 = p.cnt[segs * L], p2 = p1 + p.cnt[segs * L + 1];
    const int p3 = segs > 2 ? p2 + p.cnt[segs * L + 2] : p2, c_all = segs > 2 ? p3 + p.cnt[segs * L + 3] : p2;
    const int kk = p.b.kk, d = p.d;
    const float *qrow = p.q + q * (int64_t)d;
    const float qn = l2 ? p.b.qnorm[q] : 0.0f;
    if (STAGED) {
        for (int c4 = lane; c4 < d / 4; c4 += 64)
            *reinterpret_cast<float4 *>(&s_q[wave][4 * c4]) = *reinterpret_cast<const float4 *>(qrow + 4 * c4);
        __builtin_amdgcn_wave_barrier();
    }
    u64 best = 0;                                         // lane i: i-th best exact key so far
    int c = c_all;
    bool in_lds = false;
    if (p.prefilter && c_all > kk) {
        const float eps = band_eps_raw(p.xstat, d, l2 ? qn : p.b.qnorm[q], p.qres2[q], l2);
        const int ns = (c_all + 63) >> 6;                 // wave-uniform
        const int kept = ns <= 2 ? final_prefilter_ns<2>(list, seg_cap, p1, p2, p3, c_all, kk, eps, lane, s_surv[wave])
                       : ns == 3 ? final_prefilter_ns<3>(list, seg_cap, p1, p2, p3, c_all, kk, eps, lane, s_surv[wave])
                       : ns == 4 ? final_prefilter_ns<4>(list, seg_cap, p1, p2, p3, c_all, kk, eps, lane, s_surv[wave])
                                 : final_prefilter_ns<8>(list, seg_cap, p1, p2, p3, c_all, kk, eps, lane, s_surv[wave]);
        if (kept >= 0) { c = kept; in_lds = true; }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll 1
    for (int base = 0; base < c; base += 64) {
        const int e = base + lane;
        const bool valid = e < c;
        const int seg = (e >= p1) + (e >= p2) + (e >= p3);
        const int sbase = seg == 0 ? 0 : seg == 1 ? p1 : seg == 2 ? p2 : p3;
        const u64 old = !valid ? 0 : in_lds ? s_surv[wave][e] : list[seg * seg_cap + (e - sbase)];
        const u32 j = valid ? lemon_key_index(old) : 0u;   // idle lanes shadow row 0 (always allocated)
        const float *xrow = p.x + (int64_t)j * d;
        float dot;
        if (STAGED) dot = chain_dot_wave_q(s_q[wave], xrow, d, s_x[wave], lane, c - base < 64 ? c - base : 64);
        else { dot = 0.0f; for (int t = 0; t < d; ++t) dot = __builtin_fmaf(qrow[t], xrow[t], dot); }
        float sc = dot;
        if (l2) {
            const float dd = __builtin_fmaf(-2.0f, dot, qn + p.b.xnorm[j]);
            sc = -(dd > 0.0f ? dd : 0.0f);
        }
        const u64 key = (valid && old && sc == sc) ? lemon_make_key(sc, j) : 0;   // NaN is never selected
        const Ranked r = wave_rank_keys(best, key, 0, 0, 128, s_keys[wave], lane);
        s_best[wave][lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (best && r.r0 < kk) s_best[wave][r.r0] = best;
        if (key && r.r1 < kk) s_best[wave][r.r1] = key;
        __builtin_amdgcn_wave_barrier();
        best = s_best[wave][lane];
        __builtin_amdgcn_wave_barrier();
    }
    write_out_row(p.b, split, q, lane, lane < kk ? best : 0);
}

}  // namespace

// column pitch of the bf16 copies: the Q-stationary kernel is instantiated for 256/512/768
// (at 1024 the stationary operand alone would need all 256 accumulator registers next to the tile)
static int bf16_pitch(int d) {
    if (d <= 256) return 256;
    if (d <= 512) return 512;
    if (d <= 768) return 768;
    return (int)round_up(d, BKH);
}

static int convert_rows(const float *src, int64_t n, int d, lp16 *dst, int dpad_h, float *res2, float *hn2,
                        hipStream_t stream) {
    if (n <= 0) return LEMON_OK;
    hipLaunchKernelGGL(k_convert_bf16, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, src, n, d, dst, dpad_h,
                       res2, hn2);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

static int ensure_bf16_copy(lemon_index_t *idx, hipStream_t stream) {
    const int dpad_h = bf16_pitch(idx->d);
    idx->dpad_h = dpad_h;
    if (!idx->xh) {
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (hipMalloc(&idx->xh, (size_t)idx->cap * dpad_h * sizeof(unsigned short) + 16) != hipSuccess ||
            hipMalloc(&idx->xh_stats, (size_t)idx->cap * 2 * sizeof(float)) != hipSuccess ||
            (!idx->xn2max_dev && hipMalloc(&idx->xn2max_dev, 16) != hipSuccess)) {
            lemon_set_error("bf16 copy allocation failed");
            return LEMON_E_NOMEM;
        }
        LEMON_HIP_CHECK(hipMemsetAsync(idx->xh, 0, (size_t)idx->cap * dpad_h * sizeof(unsigned short), stream));
        idx->xh_rows = 0;
    }
    if (idx->xh_rows < idx->n) {
        const int64_t n_new = idx->n - idx->xh_rows;
        float *res2 = idx->xh_stats, *hn2 = idx->xh_stats + idx->cap;
        int rc = convert_rows(idx->x + idx->xh_rows * idx->d, n_new, idx->d,
                              reinterpret_cast<lp16 *>(idx->xh) + idx->xh_rows * dpad_h, dpad_h,
                              res2 + idx->xh_rows, hn2 + idx->xh_rows, stream);
        if (rc) return rc;
        LEMON_HIP_CHECK(hipMemsetAsync(idx->xn2max_dev, 0, 16, stream));
        hipLaunchKernelGGL(k_max_nonneg, dim3(256), dim3(256), 0, stream, idx->xnorm, idx->n, idx->xn2max_dev);
        hipLaunchKernelGGL(k_max_nonneg, dim3(256), dim3(256), 0, stream, res2, idx->n, idx->xn2max_dev + 1);
        hipLaunchKernelGGL(k_max_nonneg, dim3(256), dim3(256), 0, stream, hn2, idx->n, idx->xn2max_dev + 2);
        LEMON_HIP_CHECK(hipGetLastError());
        idx->xh_rows = idx->n;
    }
    return LEMON_OK;
}

static const int64_t QCHUNK_H = 1 << 19;

// QS2 (two query blocks per wave, 256 queries per workgroup, 64-row tiles) serves the register-resident pitches;
// LEMON_QS2=0 selects the one-block kernel (A/B aid)
static bool use_qs2() {
    const char *e = getenv("LEMON_QS2");
    return !(e && e[0] == '0');
}

template <bool l2, bool PROF>
static void launch_qs2(int kt, unsigned grid, hipStream_t stream, const ScanParamsH &p) {
    static const bool accv = [] { const char *e = getenv("LEMON_QS2_ACCV"); return !(e && e[0] == '0'); }();   // A/B aid
    switch (kt) {
        case 8:
            if (accv) hipLaunchKernelGGL((k_scan_bf16_qs2<8, 0, l2, PROF, true>), dim3(grid), dim3(NT), 0, stream, p);
            else      hipLaunchKernelGGL((k_scan_bf16_qs2<8, 0, l2, PROF, false>), dim3(grid), dim3(NT), 0, stream, p);
            break;
        default:    // (the L2 epilogue's |x|^2 loads need a few more registers: four more parked k-steps)
            if (accv) hipLaunchKernelGGL((k_scan_bf16_qs2<12, l2 ? 20 : 16, l2, PROF, true>), dim3(grid), dim3(NT), 0, stream, p);
            else      hipLaunchKernelGGL((k_scan_bf16_qs2<12, l2 ? 20 : 16, l2, PROF, false>), dim3(grid), dim3(NT), 0, stream, p);
            break;
    }
}

// QS4 = the QS2 work decomposition on v_mfma_f32_16x16x32_f16 (round 5); LEMON_QS4=0 selects QS2 (A/B aid)
static bool use_qs4() {
    const char *e = getenv("LEMON_QS4");
    return !(e && e[0] == '0');
}

// (the L2 epilogue's |x|^2 loads and |q|^2 registers do not fit next to d = 768's fragments -- 14 spilled registers with 20 parked
// steps, and 21 is what the LDS holds --, so squared-L2 at pitch 768 stays on QS2)
static bool qs4_serves(int kt, bool l2) { return kt == 8 || !l2; }

template <bool l2>
static void launch_qs4(int kt, unsigned grid, hipStream_t stream, const ScanParamsH &p) {
    if (kt == 8) hipLaunchKernelGGL((k_scan_f16_qs4<8, 0, l2>), dim3(grid), dim3(NT), 0, stream, p);
    else if (!l2) hipLaunchKernelGGL((k_scan_f16_qs4<12, 16, false>), dim3(grid), dim3(NT), 0, stream, p);
}

int lemon_search_bf16(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                      int64_t *I_dev, hipStream_t stream) {
    const int d = idx->d;
    if (idx->n == 0) return lemon_fill_empty(D_dev, I_dev, nq * k, idx->metric, stream);
    int rc = ensure_bf16_copy(idx, stream);
    if (rc) return rc;
    const int dpad_h = idx->dpad_h;
    const bool qs = dpad_h <= 768;
    // QS2 halves the panel count.  When 256-query panels alone do not fill the chip the database would be split between more
    // workgroups, and every split pays its own cold start (k ln(n/k) appends per query): measured 13.0 -> 20.1 ms at
    // 50 000 x 40 000 x 512 and 17.3 -> 21.8 ms at 131 072^2 x 256 (k = 11), against 1 648 -> 1 501 ms at 1 M x 768,
    // 142.1 -> 129.9 ms at 262 144^2 x 768 and 103.9 -> 97.8 ms at 262 144^2 x 512.  So: from 768 panels of 256 queries on
    // (the splits == 1 regime of lemon_plan_splits), pitches 512 and 768.
    const char *qs2_env = getenv("LEMON_QS2_MIN_PANELS");      // (read per call: the tests force QS2 onto small shapes with 0)
    const int qs2_min = qs2_env ? atoi(qs2_env) : 768;
    static const int cus = [] {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        return c > 0 ? c : 256;
    }();
    const int n_tiles128 = (int)((idx->n + BX - 1) / BX);
    static const bool rest_qs4 = [] { const char *e = getenv("LEMON_QS4_REST"); return !(e && e[0] == '0'); }();   // (A/B aid)
    bool prev_qs4 = false;
    int64_t cn = 0;
    for (int64_t c0 = 0; c0 < nq; c0 += cn) {
        cn = (nq - c0) < QCHUNK_H ? (nq - c0) : QCHUNK_H;
        bool qs2 = qs && use_qs2() && dpad_h >= 512 && cn >= (int64_t)qs2_min * BQ2;
        bool qs4 = qs2 && use_qs4() && qs4_serves(dpad_h / BKH, idx->metric == LEMON_METRIC_L2);
        // The ragged rest behind whole-round QS4 chunks (1 M queries: 16 960 = 67 panels of 256): the same kernel with the database
        // split between a few workgroups per panel -- the smallest split count that fills at least three quarters of the rounds it
        // takes (67 panels x 3 = 201 of 256 slots) -- instead of 133 one-block panels x 6 splits (39 ms per modality at 1 M x 768).
        int rest_splits = 0;
        if (!qs2 && prev_qs4 && c0 > 0 && rest_qs4) {
            const int panels_r = (int)((cn + BQ2 - 1) / BQ2);
            for (int sp = 1; sp <= 16 && !rest_splits; ++sp) {
                const int64_t wgs = (int64_t)panels_r * sp, rounds = (wgs + cus - 1) / cus;
                if (wgs * 4 >= rounds * cus * 3 && n_tiles128 / sp >= 64) rest_splits = sp;
            }
            if (rest_splits) qs2 = qs4 = true;
        }
        prev_qs4 = qs4;
        if (qs2 && cn < QCHUNK_H) {
            // Whole rounds first.  The chunked scan runs ONE workgroup per CU, all of equal length: 1 859 workgroups take
            // eight rounds of 256 like 2 048 do (1 M queries = 2 048 + 1 859 panels: 4.6 % of the scan spent in a quarter-full
            // last round).  So a final stretch that does not fill its last round to 80 % is cut at the last whole round; the
            // ragged rest (< 205 panels) comes back through this loop, is too small for QS2 and goes through the one-block
            // kernel with the database split between workgroups -- a few rounds of 1/splits of the scan each.
            const int64_t panels_c = (cn + BQ2 - 1) / BQ2, full = panels_c / cus * cus;
            if (full > 0 && panels_c != full && (panels_c - full) * 5 < (int64_t)cus * 4) cn = full * BQ2;
        }
        const int bqw = qs2 ? BQ2 : BQ;                     // queries per workgroup
        const int rt = qs2 ? RT2 : BX;                      // database rows per tile
        const int n_tiles = (int)((idx->n + rt - 1) / rt);
        const int64_t nq_pad = round_up(cn, bqw);
        const int panels = (int)(nq_pad / bqw);
        int splits, tiles_per_split;
        lemon_plan_splits(panels, n_tiles128, &splits, &tiles_per_split);
        if (rest_splits) {
            tiles_per_split = (n_tiles128 + rest_splits - 1) / rest_splits;
            splits = (n_tiles128 + tiles_per_split - 1) / tiles_per_split;
        }
        if (qs2) tiles_per_split *= BX / RT2;            // (the plan counts 128-row tiles)
        rc = lemon_ensure_search_ws(idx, nq_pad, splits, (int64_t)panels * splits * (bqw / BQ), dpad_h * 2, CAPH, stream);
        if (rc) return rc;
        // bf16 query panel (pad rows zero), chain norms, measured rounding residuals
        lp16 *qh = reinterpret_cast<lp16 *>(idx->ws_qp);
        float *qn = idx->ws_qnorm, *qres2 = idx->ws_qnorm + idx->ws_q, *qhn2 = idx->ws_qnorm + 2 * idx->ws_q;
        LEMON_HIP_CHECK(hipMemsetAsync(qh, 0, (size_t)nq_pad * dpad_h * sizeof(unsigned short), stream));
        LEMON_HIP_CHECK(hipMemsetAsync(idx->ws_qnorm, 0, (size_t)idx->ws_q * 3 * sizeof(float), stream));
        rc = convert_rows(q_dev + c0 * d, cn, d, qh, dpad_h, qres2, qhn2, stream);
        if (rc) return rc;
        rc = lemon_rowdot_chain(q_dev + c0 * d, q_dev + c0 * d, cn, d, qn, stream);
        if (rc) return rc;

        ScanParamsH p;
        p.b.qp = nullptr; p.b.xp = nullptr; p.b.qnorm = qn; p.b.xnorm = idx->xnorm;
        p.b.cand = idx->ws_cand; p.b.part = idx->ws_part;
        p.b.D = D_dev + c0 * k; p.b.I = I_dev + c0 * k;
        p.b.nq = cn; p.b.n = idx->n; p.b.dpad = dpad_h; p.b.kk = k; p.b.metric = idx->metric;
        p.b.n_tiles = n_tiles; p.b.tiles_per_split = tiles_per_split; p.b.splits = splits; p.b.nq_pad = nq_pad;
        p.qh = qh; p.xh = reinterpret_cast<const lp16 *>(idx->xh);
        p.q = q_dev + c0 * d; p.x = idx->x; p.qres2 = qres2; p.qhn2 = qhn2; p.xstat = idx->xn2max_dev;
        p.d = d; p.dpad_h = dpad_h; p.phase_dbg = nullptr;
        rc = lemon_parse_ablate("k_scan_bf16_qs", 1 | 4 | 8, &p.ablate);   // bit 1 (value 2) does not exist: see ScanParamsH
        if (rc) return rc;
        static const int refresh = [] { const char *e = getenv("LEMON_REFRESH"); return e && atoi(e) > 0 ? atoi(e) : REFRESH; }();
        p.b.stale = refresh;                  // new candidates per query that trigger a light compaction (tuning knob)
        const unsigned grid = (unsigned)(panels * splits);
        // database chunks sized for the Infinity Cache (the chunk is re-read by every query panel)
        int chunk_tiles = n_tiles;
        if (qs && splits == 1) {
            const char *env = getenv("LEMON_CHUNK_MB");
            const double mb = env ? atof(env) : 64.0;
            if (mb > 0) {
                chunk_tiles = (int)(mb * 1048576.0 / ((double)rt * dpad_h * 2));
                if (chunk_tiles < 8) chunk_tiles = 8;
            }
            if (chunk_tiles > n_tiles) chunk_tiles = n_tiles;
        }
        // per-lane state carried between chunk launches (splits == 1) + the half-list counts handed to k_bf16_final
        const int64_t state_elems = (qs && splits == 1) ? (int64_t)grid * NT * (qs4 ? 16 : qs2 ? 8 : 4) : 0;
        const int64_t cnt_elems = qs ? (int64_t)grid * bqw * (qs4 ? 4 : 2) : 0;
        if (state_elems + cnt_elems > idx->ws_state_elems) {
            LEMON_HIP_CHECK(hipStreamSynchronize(stream));
            if (idx->ws_state) (void)hipFree(idx->ws_state);
            idx->ws_state = nullptr; idx->ws_state_elems = 0;
            if (hipMalloc(&idx->ws_state, (size_t)(state_elems + cnt_elems) * sizeof(float)) != hipSuccess) {
                lemon_set_error("scan state allocation failed");
                return LEMON_E_NOMEM;
            }
            idx->ws_state_elems = state_elems + cnt_elems;
        }
        p.state = idx->ws_state;
        p.cnt = reinterpret_cast<int *>(idx->ws_state + state_elems);
        const bool l2m_ = idx->metric == LEMON_METRIC_L2;
        FinalParams fp;
        fp.b = p.b; fp.q = p.q; fp.x = p.x; fp.cnt = p.cnt; fp.d = d; fp.rows_per_wg = bqw; fp.segs = qs4 ? 4 : 2; fp.n_lists = (int64_t)grid * bqw;
        fp.qres2 = qres2; fp.xstat = idx->xn2max_dev;
        static const int prefilter = [] { const char *e = getenv("LEMON_FINAL_PREFILTER"); return !(e && e[0] == '0'); }();   // (A/B aid)
        fp.prefilter = prefilter;
        auto launch_final = [&]() {     // exact re-scoring + exact top-k of every (query, split) list, one wave each
            const unsigned fg = (unsigned)((fp.n_lists + 3) / 4);
            const bool staged = (d % 4) == 0 && d <= 1024;
            if (l2m_) { if (staged) hipLaunchKernelGGL((k_bf16_final<true, true>), dim3(fg), dim3(256), 0, stream, fp);
                        else        hipLaunchKernelGGL((k_bf16_final<true, false>), dim3(fg), dim3(256), 0, stream, fp); }
            else      { if (staged) hipLaunchKernelGGL((k_bf16_final<false, true>), dim3(fg), dim3(256), 0, stream, fp);
                        else        hipLaunchKernelGGL((k_bf16_final<false, false>), dim3(fg), dim3(256), 0, stream, fp); }
        };
        const bool l2m = idx->metric == LEMON_METRIC_L2;
        for (int t0 = 0; t0 < n_tiles; t0 += chunk_tiles) {
            const int t1 = (t0 + chunk_tiles < n_tiles) ? t0 + chunk_tiles : n_tiles;
            p.chunk_t0 = t0; p.chunk_t1 = t1; p.first_chunk = (t0 == 0); p.last_chunk = (t1 == n_tiles);
            const double rows = (double)(t1 - t0) * rt < (double)idx->n - (double)t0 * rt ? (double)(t1 - t0) * rt
                                                                                            : (double)idx->n - (double)t0 * rt;
            const double flops = 2.0 * (double)cn * rows * (double)d;
            const double bytes = 2.0 * d * ((double)nq_pad / BQ * rows) + (p.last_chunk ? 2.0 * d * cn + 12.0 * k * (double)cn : 0.0);
            LemonProfScope prof(idx, stream, flops, bytes);
            if (qs4) {
#ifdef LEMON_QS4_PHASES
                static unsigned long long *dbg4 = nullptr;
                if (!dbg4) { (void)hipMalloc(&dbg4, 64); }
                (void)hipMemsetAsync(dbg4, 0, 64, stream);
                p.phase_dbg = dbg4;
#endif
                if (l2m) launch_qs4<true>(dpad_h / BKH, grid, stream, p);
                else     launch_qs4<false>(dpad_h / BKH, grid, stream, p);
#ifdef LEMON_QS4_PHASES
                (void)hipStreamSynchronize(stream);
                unsigned long long h4[8];
                (void)hipMemcpy(h4, dbg4, 64, hipMemcpyDeviceToHost);
                fprintf(stderr, "[qs4 phases] grid=%u tiles/wg=%.0f loop=%.4g cyc/wg  step0=%.1f%% (%.0f cyc/tile) maintain=%.1f%% sync=%.1f%% (%.0f cyc/tile)  tiles with slow path %.1f%%, with compaction %.2f%%\n",
                        grid, (double)h4[4] / grid, (double)h4[0] / grid, 100.0 * h4[1] / h4[0], (double)h4[1] / (double)h4[4], 100.0 * h4[2] / h4[0],
                        100.0 * h4[3] / h4[0], (double)h4[3] / (double)h4[4], 100.0 * h4[5] / (double)h4[4], 100.0 * h4[6] / (double)h4[4]);
#endif
                if (p.last_chunk) launch_final();
                continue;
            }
            if (qs && dpad_h / BKH == 12 && !l2m && getenv("LEMON_PHASE_PROF")) {   // diagnostic build: phase cycle sums
                static unsigned long long *dbg = nullptr;
                if (!dbg) { (void)hipMalloc(&dbg, 64); (void)hipMemset(dbg, 0, 64); }
                p.phase_dbg = dbg;
                if (qs2) launch_qs2<false, true>(12, grid, stream, p);
                else hipLaunchKernelGGL((k_scan_bf16_qs<12, false, true>), dim3(grid), dim3(NT), 0, stream, p);
                (void)hipStreamSynchronize(stream);
                unsigned long long h[8]; (void)hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost);

                const double tot = (double)(h[0] + h[1] + h[2] + h[3]);
                fprintf(stderr, "[phase] grid=%u loop=%.1f%% epilogue=%.1f%% sync=%.1f%% maintain=%.1f%% total=%.3g cyc/WG=%.3g\n",
                        grid, 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, tot, tot / grid);
                (void)hipMemset(dbg, 0, 64);
                if (p.last_chunk) launch_final();
                continue;
            }
            if (qs2) {
                if (l2m) launch_qs2<true, false>(dpad_h / BKH, grid, stream, p);
                else     launch_qs2<false, false>(dpad_h / BKH, grid, stream, p);
                if (p.last_chunk) launch_final();
                continue;
            }
            switch (qs ? dpad_h / BKH : 0) {
                case 4:
                    if (l2m) hipLaunchKernelGGL((k_scan_bf16_qs<4, true, false>), dim3(grid), dim3(NT), 0, stream, p);
                    else     hipLaunchKernelGGL((k_scan_bf16_qs<4, false, false>), dim3(grid), dim3(NT), 0, stream, p);
                    break;
                case 8:
                    if (l2m) hipLaunchKernelGGL((k_scan_bf16_qs<8, true, false>), dim3(grid), dim3(NT), 0, stream, p);
                    else     hipLaunchKernelGGL((k_scan_bf16_qs<8, false, false>), dim3(grid), dim3(NT), 0, stream, p);
                    break;
                case 12:
                    if (l2m) hipLaunchKernelGGL((k_scan_bf16_qs<12, true, false>), dim3(grid), dim3(NT), 0, stream, p);
                    else     hipLaunchKernelGGL((k_scan_bf16_qs<12, false, false>), dim3(grid), dim3(NT), 0, stream, p);
                    break;
                default: hipLaunchKernelGGL(k_scan_bf16, dim3(grid), dim3(NT), 0, stream, p); break;
            }
            if (qs && p.last_chunk) launch_final();
        }
        LEMON_HIP_CHECK(hipGetLastError());
        if (splits > 1) {
            rc = lemon_launch_merge(idx->ws_part, splits, nullptr, nq_pad, cn, k, idx->metric, p.b.D, p.b.I, stream);
            if (rc) return rc;
        }
        idx->last.algo = LEMON_ALGO_BF16_FILTER;
        if (c0 == 0) {                                  // (the first chunk is the largest: its geometry is what gets reported)
            idx->last.grid = (int)grid; idx->last.block = NT;
            idx->last.query_panel = bqw; idx->last.db_splits = splits;
        }
    }
    idx->last.nq = nq; idx->last.n = idx->n; idx->last.d = d; idx->last.k = k;
    return LEMON_OK;
}
