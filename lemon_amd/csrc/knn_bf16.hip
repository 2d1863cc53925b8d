// knn_bf16.hip -- LEMON_ALGO_BF16_FILTER: bf16 MFMA filter scan + exact float32 re-rank.
//
// Same skeleton as knn_f32.hip (a workgroup owns 128 queries and streams database tiles through
// LDS) but the tile product runs on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate, half the
// staged bytes) over bf16 (round-to-nearest-even) copies of Q and X.  The approximate score s~ is
// only a FILTER:
//     |s~(q,x) - s(q,x)| <= eps(q) := C_REL(d) * ||q|| * max_j ||x_j||        (bound derived below)
// where s is the float32 fmaf-chain score of the numeric contract.  With T = the exact score of
// the query's current k-th best (rows are visited in ascending index, so a later row only matters
// if s > T strictly), every row that can still enter the top-k satisfies s~ > T - eps: those rows
// are appended (index only) to the query's pending list.  When a list could overflow, one
// wavefront re-scores its pending rows EXACTLY (one lane per row walks the float32 fmaf chain over
// the original data), merges them with the exact top-k by rank-select and tightens T.  The result
// is bit-identical to LEMON_ALGO_F32_MFMA and to the CPU oracle; no fallback path is needed
// because the band is a proof, not a heuristic.
//
// eps: bf16 RNE has relative error <= 2^-8 per element, so for exact arithmetic on rounded inputs
// |sum qh_i xh_i - sum q_i x_i| <= (2*2^-8 + 2^-16) sum |q_i x_i| <= (2^-7 + 2^-16) ||q|| ||x||
// (Cauchy-Schwarz).  Float32 accumulation adds at most ~d*2^-24 sum|q_i x_i| on either side (the
// MFMA's and the chain's); we budget 8*d*2^-24 for both, inflate the norms by 2^-10 for their own
// rounding, and add an absolute 1e-30 for subnormal inputs.
#include "knn_common.hpp"

using namespace lemon_knn;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int PEND_CAP = (CAP - LEMON_MAX_K) * 2;  // 384 u32 row indices behind the 64 exact keys
constexpr int BKH = 64;                            // bf16 k-slice per LDS stage (128 B rows, like fp32 BK=32)

// f32 [n,d] -> bf16 [*, dpad_h] (RNE, zero padded columns); one thread per 8 outputs
__global__ __launch_bounds__(256) void k_convert_bf16(const float *__restrict__ src, int64_t n, int d,
                                                      __bf16 *__restrict__ dst, int dpad_h) {
    const int groups = dpad_h / 8;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * groups) return;
    const int64_t r = t / groups;
    const int u = (int)(t % groups);
    const float *s = src + r * (int64_t)d + 8 * u;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)((8 * u + e < d) ? s[e] : 0.0f);
    *reinterpret_cast<bf16x8 *>(dst + r * (int64_t)dpad_h + 8 * u) = o;
}

// max of non-negative floats through their bit patterns
__global__ __launch_bounds__(256) void k_max_nonneg(const float *__restrict__ v, int64_t n, unsigned *__restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = v[i];
        m = (x > m || x != x) ? x : m;      // NaN/Inf propagate: the band then admits everything
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
    const __bf16 *qh, *xh;     // [nq_pad, dpad_h], [n_pad, dpad_h]
    const float *q, *x;        // originals, row-major [nq, d], [n, d]
    const unsigned *xn2max;    // device scalar: max_j dot(x_j,x_j) as float bits
    int d, dpad_h;
    float c_rel;
};

// exact score of the numeric contract for (query row, db row j); key to MAXIMISE
__device__ __forceinline__ float exact_score(const float *__restrict__ q, const float *__restrict__ x, int d,
                                             bool l2, float qn, float xn) {
    float acc = 0.0f;
    if ((d & 3) == 0) {
        const float4 *q4 = reinterpret_cast<const float4 *>(q);
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
#pragma unroll 4
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

// re-score the pending rows of one query exactly, merge with its exact top-k, tighten the thresholds
__device__ __forceinline__ void compact_row_h(const ScanParamsH &p, u64 *__restrict__ list, int row, int64_t q,
                                              int lane, int *s_cnt, int *s_kept, float *s_thr_lo, const float *s_eps,
                                              const float *s_qn, u64 *__restrict__ sk, u64 *__restrict__ sb) {
    const int n_p = __builtin_amdgcn_readfirstlane(s_cnt[row]);
    const int kept = __builtin_amdgcn_readfirstlane(s_kept[row]);
    const int kk = p.b.kk;
    const u32 *pend = reinterpret_cast<const u32 *>(list + LEMON_MAX_K);
    const float *qrow = p.q + q * (int64_t)p.d;
    const float qn = s_qn[row];
    const bool l2 = p.b.metric == LEMON_METRIC_L2;
    u64 best = (lane < kept) ? list[lane] : 0;
#pragma unroll 1
    for (int base = 0; base < n_p; base += 64) {     // one exact chain per lane, then a 128-key rank merge
        const int e = base + lane;
        u64 key = 0;
        if (e < n_p) {
            const u32 j = pend[e];
            const float *xrow = p.x + (int64_t)j * p.d;
            const float s = exact_score(qrow, xrow, p.d, l2, qn, l2 ? p.b.xnorm[j] : 0.0f);
            key = (s == s) ? lemon_make_key(s, j) : 0;           // NaN is never selected (as in the fp32 scan)
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
        s_cnt[row] = 0;
        s_kept[row] = have;
        if (have == kk) {
            const float T = lemon_key_score(kth);
            // admit s~ > T - eps; round the bound DOWN (a few ulps of slack never hurts correctness)
            const float lo = T - s_eps[row];
            s_thr_lo[row] = lo - fabsf(lo) * 2.4e-7f - 1e-37f;
        }
    }
}

// filter one 32x32 accumulator tile: append the row index of every element above its row's bound
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
                const int slot = atomicAdd(&s_cnt[row], 1);
                reinterpret_cast<u32 *>(cand_panel + (int64_t)row * CAP + LEMON_MAX_K)[slot] = j;
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
    __shared__ float s_eps[BQ];
    __shared__ int s_cnt[BQ];
    __shared__ int s_kept[BQ];
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
    const int metric = p.b.metric;

    if (tid < BQ) {
        const bool valid = q0 + tid < p.b.nq;
        const float qn = p.b.qnorm[q0 + tid];
        const float xn2 = __uint_as_float(*p.xn2max);
        float eps = p.c_rel * sqrtf(qn) * sqrtf(xn2) * 1.002f + 1e-30f;
        if (metric == LEMON_METRIC_L2) eps = 2.0f * eps + 4.8e-7f * (qn + xn2);
        s_eps[tid] = eps;
        s_qn[tid] = qn;
        s_thr_lo[tid] = valid ? -INFINITY : INFINITY;
        s_cnt[tid] = 0;
        s_kept[tid] = 0;
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

    u64 *cand_panel = p.b.cand + (int64_t)blockIdx.x * BQ * CAP;
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
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc11, 0, 0, 0);
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
            const bool last = (it + 1 == total);
            for (int r = 0; r < 32; ++r) {
                const int row = 32 * wave + r;
                const int c = s_cnt[row];
                const bool warm = (s_thr_lo[row] == -INFINITY) && (c + s_kept[row] >= p.b.kk);
                if (c > 0 && (last || c > PEND_CAP - BX || warm))
                    compact_row_h(p, cand_panel + (int64_t)row * CAP, row, q0 + row, lane, s_cnt, s_kept, s_thr_lo,
                                  s_eps, s_qn, s_keys[wave], s_best[wave]);
            }
            __syncthreads();
        }
        kt = kt_n; jl = jl_n;
    }

    for (int r = 0; r < 32; ++r) {
        const int row = 32 * wave + r;
        const int64_t q = q0 + row;
        if (q >= p.b.nq) continue;
        const int kept = s_kept[row];
        const u64 key = (lane < kept && lane < p.b.kk) ? cand_panel[(int64_t)row * CAP + lane] : 0;
        write_out_row(p.b, split, q, lane, key);
    }
}

}  // namespace

static int ensure_bf16_copy(lemon_index_t *idx, hipStream_t stream) {
    const int dpad_h = (int)round_up(idx->d, BKH);
    if (!idx->xh) {
        // cap rows + one extra tile of zero rows so the last tile never reads past the allocation
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (hipMalloc(&idx->xh, (size_t)idx->cap * dpad_h * sizeof(unsigned short) + 16) != hipSuccess ||
            (!idx->xn2max_dev && hipMalloc(&idx->xn2max_dev, 16) != hipSuccess)) {
            lemon_set_error("bf16 copy allocation failed");
            return LEMON_E_NOMEM;
        }
        LEMON_HIP_CHECK(hipMemsetAsync(idx->xh, 0, (size_t)idx->cap * dpad_h * sizeof(unsigned short), stream));
        idx->xh_rows = 0;
    }
    if (idx->xh_rows < idx->n) {
        const int64_t n_new = idx->n - idx->xh_rows;
        const int64_t threads = n_new * (dpad_h / 8);
        hipLaunchKernelGGL(k_convert_bf16, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                           idx->x + idx->xh_rows * idx->d, n_new, idx->d,
                           reinterpret_cast<__bf16 *>(idx->xh) + idx->xh_rows * dpad_h, dpad_h);
        LEMON_HIP_CHECK(hipGetLastError());
        LEMON_HIP_CHECK(hipMemsetAsync(idx->xn2max_dev, 0, 4, stream));
        hipLaunchKernelGGL(k_max_nonneg, dim3(256), dim3(256), 0, stream, idx->xnorm, idx->n, idx->xn2max_dev);
        LEMON_HIP_CHECK(hipGetLastError());
        idx->xh_rows = idx->n;
    }
    return LEMON_OK;
}

static const int64_t QCHUNK_H = 1 << 19;

int lemon_search_bf16(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                      int64_t *I_dev, hipStream_t stream) {
    const int d = idx->d;
    if (idx->n == 0) return lemon_fill_empty(D_dev, I_dev, nq * k, idx->metric, stream);
    int rc = ensure_bf16_copy(idx, stream);
    if (rc) return rc;
    const int dpad_h = (int)round_up(d, BKH);
    const int n_tiles = (int)((idx->n + BX - 1) / BX);
    for (int64_t c0 = 0; c0 < nq; c0 += QCHUNK_H) {
        const int64_t cn = (nq - c0) < QCHUNK_H ? (nq - c0) : QCHUNK_H;
        const int64_t nq_pad = round_up(cn, BQ);
        const int panels = (int)(nq_pad / BQ);
        int splits, tiles_per_split;
        lemon_plan_splits(panels, n_tiles, &splits, &tiles_per_split);
        rc = lemon_ensure_search_ws(idx, nq_pad, splits, dpad_h * 2, stream);
        if (rc) return rc;
        // bf16 query panel (pad rows zero) + chain norms (band + L2)
        __bf16 *qh = reinterpret_cast<__bf16 *>(idx->ws_qp);
        LEMON_HIP_CHECK(hipMemsetAsync(qh, 0, (size_t)nq_pad * dpad_h * sizeof(unsigned short), stream));
        {
            const int64_t threads = cn * (dpad_h / 8);
            hipLaunchKernelGGL(k_convert_bf16, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                               q_dev + c0 * d, cn, d, qh, dpad_h);
            LEMON_HIP_CHECK(hipGetLastError());
        }
        LEMON_HIP_CHECK(hipMemsetAsync(idx->ws_qnorm, 0, (size_t)nq_pad * sizeof(float), stream));
        rc = lemon_rowdot_chain(q_dev + c0 * d, q_dev + c0 * d, cn, d, idx->ws_qnorm, stream);
        if (rc) return rc;

        ScanParamsH p;
        p.b.qp = nullptr; p.b.xp = nullptr; p.b.qnorm = idx->ws_qnorm; p.b.xnorm = idx->xnorm;
        p.b.cand = idx->ws_cand; p.b.part = idx->ws_part;
        p.b.D = D_dev + c0 * k; p.b.I = I_dev + c0 * k;
        p.b.nq = cn; p.b.n = idx->n; p.b.dpad = dpad_h; p.b.kk = k; p.b.metric = idx->metric;
        p.b.n_tiles = n_tiles; p.b.tiles_per_split = tiles_per_split; p.b.splits = splits; p.b.nq_pad = nq_pad;
        p.qh = qh; p.xh = reinterpret_cast<const __bf16 *>(idx->xh);
        p.q = q_dev + c0 * d; p.x = idx->x; p.xn2max = idx->xn2max_dev;
        p.d = d; p.dpad_h = dpad_h;
        p.c_rel = 0.0078125f + 1.6e-5f + 8.0f * (float)d * 5.9604645e-8f;   // 2^-7 + 2^-16 + 8 d 2^-24
        const unsigned grid = (unsigned)(panels * splits);
        {
            const double flops = 2.0 * (double)cn * (double)idx->n * (double)d;
            const double bytes = 2.0 * d * ((double)cn + (double)panels * (double)idx->n) + 12.0 * k * (double)cn;
            LemonProfScope prof(idx, stream, flops, bytes);
            hipLaunchKernelGGL(k_scan_bf16, dim3(grid), dim3(NT), 0, stream, p);
        }
        LEMON_HIP_CHECK(hipGetLastError());
        if (splits > 1) {
            rc = lemon_launch_merge(idx->ws_part, splits, nq_pad, cn, k, idx->metric, p.b.D, p.b.I, stream);
            if (rc) return rc;
        }
        idx->last.algo = LEMON_ALGO_BF16_FILTER;
        idx->last.grid = (int)grid; idx->last.block = NT;
        idx->last.query_panel = BQ; idx->last.db_splits = splits;
    }
    idx->last.nq = nq; idx->last.n = idx->n; idx->last.d = d; idx->last.k = k;
    return LEMON_OK;
}
