// api.hip -- C ABI glue: index lifetime, search dispatch, multimodal-neighbour finalisation.
#include "common.hpp"
#include <stdlib.h>
#include <math.h>

static thread_local char g_err[512] = "";

void lemon_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *lemon_last_error(void) { return g_err; }
extern "C" int lemon_version(int *major, int *minor) {
    if (major) *major = 0;
    if (minor) *minor = 1;
    return LEMON_OK;
}

int lemon_permute_rows(const float *src, int64_t n, int d, float *dst, int dpad, hipStream_t s);
int lemon_search_f32(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                     int64_t *I_dev, hipStream_t stream);
int lemon_search_bf16(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                      int64_t *I_dev, hipStream_t stream);

static inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---- index lifetime ---------------------------------------------------------------------
extern "C" int lemon_index_create(int metric, int d, lemon_index_t **out) {
    LEMON_REQUIRE(out != nullptr, "out pointer");
    LEMON_REQUIRE(metric == LEMON_METRIC_IP || metric == LEMON_METRIC_L2, "metric");
    LEMON_REQUIRE(d > 0 && d <= 65536, "0 < d <= 65536");
    lemon_index_t *idx = (lemon_index_t *)calloc(1, sizeof(lemon_index_t));
    if (!idx) { lemon_set_error("host allocation failed"); return LEMON_E_NOMEM; }
    idx->metric = metric;
    idx->d = d;
    idx->dpad = (int)round_up64(d, 64);   // even number of 32-wide k-slices (fp32 scan's two-set prefetch)
    idx->algo = LEMON_ALGO_AUTO;
    {   // query de-duplication: on by default, LEMON_QUERY_DEDUP=0 turns it off process-wide
        const char *e = getenv("LEMON_QUERY_DEDUP");
        idx->qdedup = (e && e[0] == '0') ? 0 : 1;
    }
    idx->prof_events = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
    if (hipGetDevice(&idx->device) != hipSuccess) {
        delete idx->prof_events;
        free(idx);
        lemon_set_error("hipGetDevice failed: no HIP device available");
        return LEMON_E_HIP;
    }
    *out = idx;
    return LEMON_OK;
}

extern "C" int lemon_index_free(lemon_index_t *idx) {
    if (!idx) return LEMON_OK;
    void *ptrs[] = {idx->x, idx->xp, idx->xnorm, idx->xh, idx->xh_stats, idx->xn2max_dev, idx->ws_qp, idx->ws_qnorm,
                    idx->ws_cand, idx->ws_part, idx->ws_state, idx->ws_D, idx->ws_I, idx->ws_dd, idx->ws_ddq,
                    idx->plan_slots[0].dev, idx->plan_slots[1].dev, idx->plan_slots[2].dev, idx->plan_slots[3].dev};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &e : *idx->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete idx->prof_events;
    for (auto &slot : idx->plan_slots) delete slot.host;
    free(idx);
    return LEMON_OK;
}

extern "C" int64_t lemon_index_ntotal(const lemon_index_t *idx) { return idx ? idx->n : -1; }
extern "C" int lemon_index_dim(const lemon_index_t *idx) { return idx ? idx->d : -1; }
extern "C" const float *lemon_index_data(const lemon_index_t *idx) { return idx ? idx->x : nullptr; }

extern "C" int lemon_index_set_query_dedup(lemon_index_t *idx, int enabled) {
    LEMON_REQUIRE(idx != nullptr, "index handle");
    idx->qdedup = enabled ? 1 : 0;
    return LEMON_OK;
}

extern "C" int lemon_index_set_algo(lemon_index_t *idx, int algo) {
    LEMON_REQUIRE(idx != nullptr, "index handle");
    LEMON_REQUIRE(algo >= LEMON_ALGO_AUTO && algo <= LEMON_ALGO_BF16_FILTER, "algo");
    idx->algo = algo;
    return LEMON_OK;
}

extern "C" int lemon_index_last_search_info(const lemon_index_t *idx, lemon_search_info_t *out) {
    LEMON_REQUIRE(idx && out, "null pointer");
    *out = idx->last;
    return LEMON_OK;
}

extern "C" int lemon_index_set_profiling(lemon_index_t *idx, int enabled) {
    LEMON_REQUIRE(idx != nullptr, "index handle");
    idx->profiling = enabled ? 1 : 0;
    return LEMON_OK;
}

extern "C" int lemon_index_profile_read(lemon_index_t *idx, int64_t *launches, double *kernel_ms,
                                        double *algo_flops, double *algo_bytes) {
    LEMON_REQUIRE(idx != nullptr, "index handle");
    double ms = 0.0;
    for (auto &e : *idx->prof_events) {
        LEMON_HIP_CHECK(hipEventSynchronize(e.second));
        float t = 0.0f;
        LEMON_HIP_CHECK(hipEventElapsedTime(&t, e.first, e.second));
        ms += t;
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    if (launches) *launches = (int64_t)idx->prof_events->size();
    if (kernel_ms) *kernel_ms = ms;
    if (algo_flops) *algo_flops = idx->prof_flops;
    if (algo_bytes) *algo_bytes = idx->prof_bytes;
    idx->prof_events->clear();
    idx->prof_flops = 0.0; idx->prof_bytes = 0.0;
    return LEMON_OK;
}

extern "C" int lemon_index_add(lemon_index_t *idx, const float *x_dev, int64_t n, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LEMON_REQUIRE(idx != nullptr, "index handle");
    LEMON_REQUIRE(n >= 0, "n >= 0");
    if (n == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev != nullptr, "x_dev");
    LEMON_REQUIRE(idx->n + n < (int64_t)0xfffffff0u, "ntotal < 2^32");
    const int d = idx->d, dpad = idx->dpad;
    const int64_t need = idx->n + n;
    if (need > idx->cap) {
        // grow geometrically, rows padded to whole 128-row tiles (pad rows are zero)
        int64_t cap = idx->cap ? idx->cap : 0;
        int64_t want = cap * 2 > need ? cap * 2 : need;
        if (idx->cap == 0) want = need;
        want = round_up64(want, 128);
        float *nx = nullptr, *nxp = nullptr, *nxn = nullptr;
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (hipMalloc(&nx, (size_t)want * d * sizeof(float)) != hipSuccess ||
            hipMalloc(&nxp, (size_t)want * dpad * sizeof(float)) != hipSuccess ||
            hipMalloc(&nxn, (size_t)want * sizeof(float)) != hipSuccess) {
            if (nx) (void)hipFree(nx);
            if (nxp) (void)hipFree(nxp);
            if (nxn) (void)hipFree(nxn);
            lemon_set_error("index storage allocation failed (%lld rows x %d)", (long long)want, d);
            return LEMON_E_NOMEM;
        }
        LEMON_HIP_CHECK(hipMemsetAsync(nxp, 0, (size_t)want * dpad * sizeof(float), stream));
        LEMON_HIP_CHECK(hipMemsetAsync(nxn, 0, (size_t)want * sizeof(float), stream));
        if (idx->n > 0) {
            LEMON_HIP_CHECK(hipMemcpyAsync(nx, idx->x, (size_t)idx->n * d * sizeof(float),
                                           hipMemcpyDeviceToDevice, stream));
            LEMON_HIP_CHECK(hipMemcpyAsync(nxp, idx->xp, (size_t)idx->n * dpad * sizeof(float),
                                           hipMemcpyDeviceToDevice, stream));
            LEMON_HIP_CHECK(hipMemcpyAsync(nxn, idx->xnorm, (size_t)idx->n * sizeof(float),
                                           hipMemcpyDeviceToDevice, stream));
        }
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (idx->x) (void)hipFree(idx->x);
        if (idx->xp) (void)hipFree(idx->xp);
        if (idx->xnorm) (void)hipFree(idx->xnorm);
        if (idx->xh) { (void)hipFree(idx->xh); idx->xh = nullptr; }
        if (idx->xh_stats) { (void)hipFree(idx->xh_stats); idx->xh_stats = nullptr; }
        idx->xh_rows = 0;
        idx->x = nx; idx->xp = nxp; idx->xnorm = nxn; idx->cap = want;
    }
    LEMON_HIP_CHECK(hipMemcpyAsync(idx->x + idx->n * d, x_dev, (size_t)n * d * sizeof(float),
                                   hipMemcpyDeviceToDevice, stream));
    int rc = lemon_permute_rows(x_dev, n, d, idx->xp + idx->n * dpad, dpad, stream);
    if (rc) return rc;
    rc = lemon_rowdot_chain(x_dev, x_dev, n, d, idx->xnorm + idx->n, stream);
    if (rc) return rc;
    idx->n = need;
    return LEMON_OK;
}

// ---- LEMON_ALGO_AUTO ----------------------------------------------------------------------
// The 16-bit filter scan (knn_bf16.hip; fp16 operands) is ~5x faster than the fp32 scan on large, "spread out" data, but its work
// grows with the number of database rows whose score lies within the rounding band of a query's
// k-th best: exact duplicates (class prompts: SURVEY 0.9) or tightly concentrated embeddings put
// hundreds of rows there and every one of them must be re-scored exactly.  AUTO therefore probes:
// 64 strided queries against 2048 strided database rows with the exact fp32 scan; if the typical
// probe query sees >= 4 sampled rows within the band of its best one (i.e. >= 4*n/2048 rows at full
// scale), the data is band-crowded and the fp32 scan is used.  One small search + one host sync, done
// once per index content (cached until the next add).
static int lemon_auto_choose(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, hipStream_t stream) {
    (void)k;
    const int S = 2048, P = 64, KP = 16;
    const int d = idx->d;
    if (d > 768 || idx->n < 65536 || (double)nq * (double)idx->n < 8.0e9) return LEMON_ALGO_F32_MFMA;
    if (idx->auto_n == idx->n && idx->auto_algo) return idx->auto_algo;
    int choice = LEMON_ALGO_F32_MFMA;
    float *xs = nullptr, *qs = nullptr, *Dp = nullptr, *qn = nullptr;
    int64_t *Ip = nullptr;
    lemon_index_t *probe = nullptr;
    const int64_t xstride = idx->n / S, qstride = nq >= P ? nq / P : 1;
    const int np = nq >= P ? P : (int)nq;
    do {
        if (hipMalloc(&xs, (size_t)S * d * 4) != hipSuccess || hipMalloc(&qs, (size_t)P * d * 4) != hipSuccess ||
            hipMalloc(&Dp, (size_t)P * KP * 4) != hipSuccess || hipMalloc(&Ip, (size_t)P * KP * 8) != hipSuccess ||
            hipMalloc(&qn, (size_t)P * 4) != hipSuccess) break;
        if (hipMemcpy2DAsync(xs, (size_t)d * 4, idx->x, (size_t)xstride * d * 4, (size_t)d * 4, S,
                             hipMemcpyDeviceToDevice, stream) != hipSuccess) break;
        if (hipMemcpy2DAsync(qs, (size_t)d * 4, q_dev, (size_t)qstride * d * 4, (size_t)d * 4, np,
                             hipMemcpyDeviceToDevice, stream) != hipSuccess) break;
        if (lemon_index_create(idx->metric, d, &probe) != LEMON_OK) break;
        probe->algo = LEMON_ALGO_F32_MFMA;
        if (lemon_index_add(probe, xs, S, stream) != LEMON_OK) break;
        if (lemon_search_internal(probe, qs, np, KP, Dp, Ip, stream) != LEMON_OK) break;
        if (lemon_rowdot_chain(qs, qs, np, d, qn, stream) != LEMON_OK) break;
        float hD[P * KP], hq[P], hx[S];
        if (hipMemcpyAsync(hD, Dp, (size_t)np * KP * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) break;
        if (hipMemcpyAsync(hq, qn, (size_t)np * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) break;
        if (hipMemcpyAsync(hx, probe->xnorm, (size_t)S * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) break;
        if (hipStreamSynchronize(stream) != hipSuccess) break;
        float xn2max = 0.0f;
        for (int i = 0; i < S; ++i) xn2max = hx[i] > xn2max ? hx[i] : xn2max;
        int crowded = 0;
        for (int p = 0; p < np; ++p) {
            // a-priori size of the scan's band: two fp16 roundings (2^-12 / sqrt(3) of the norms each, measured 1.4e-4) + the
            // fp32 summation term of band_eps (knn_bf16.hip); the scan itself uses the MEASURED residuals
            const float eps = (0.0004f + 3.0f * (float)d * 5.9604645e-8f) * sqrtf(hq[p] > 0 ? hq[p] : 0.0f) * sqrtf(xn2max);
            const float band = (idx->metric == LEMON_METRIC_L2) ? 4.0f * eps : 2.0f * eps;
            int c = 0;
            for (int j = 1; j < KP; ++j) {
                const float gap = (idx->metric == LEMON_METRIC_L2) ? hD[p * KP + j] - hD[p * KP] : hD[p * KP] - hD[p * KP + j];
                if (gap < band) ++c;
            }
            if (c >= 4) ++crowded;
        }
        choice = (2 * crowded >= np) ? LEMON_ALGO_F32_MFMA : LEMON_ALGO_BF16_FILTER;
    } while (0);
    if (probe) lemon_index_free(probe);
    void *ptrs[] = {xs, qs, Dp, Ip, qn};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    idx->auto_algo = choice;
    idx->auto_n = idx->n;
    return choice;
}

int lemon_search_internal(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                          int64_t *I_dev, hipStream_t stream) {
    LEMON_REQUIRE(idx != nullptr, "index handle");
    LEMON_REQUIRE(nq >= 0, "nq >= 0");
    LEMON_REQUIRE(k >= 1 && k <= LEMON_MAX_K_DEEP, "1 <= k <= LEMON_MAX_K_DEEP");
    if (nq == 0) return LEMON_OK;
    LEMON_REQUIRE(q_dev && D_dev && I_dev, "null pointer");
    // identical query rows (class prompts: 50 000 text queries, 100 distinct rows on CIFAR-100) are searched once
    if (idx->qdedup && nq >= LEMON_DEDUP_MIN_NQ && nq < ((int64_t)1 << 31) && idx->n > 0) {
        int64_t U = 0;
        const int *rep = nullptr, *grp = nullptr;
        int rc = lemon_dedup_queries(idx, q_dev, nq, stream, &U, &rep, &grp);
        if (rc) return rc;
        if (U * 2 <= nq) {
            const size_t qb = ((size_t)U * idx->d * 4 + 255) & ~(size_t)255, db = ((size_t)U * k * 4 + 255) & ~(size_t)255;
            const size_t need = qb + db + (size_t)U * k * 8;
            if (need > idx->ws_ddq_bytes) {
                LEMON_HIP_CHECK(hipStreamSynchronize(stream));
                if (idx->ws_ddq) (void)hipFree(idx->ws_ddq);
                idx->ws_ddq = nullptr; idx->ws_ddq_bytes = 0;
                if (hipMalloc((void **)&idx->ws_ddq, need) != hipSuccess) { lemon_set_error("dedup result workspace allocation failed"); return LEMON_E_NOMEM; }
                idx->ws_ddq_bytes = need;
            }
            float *qrep = idx->ws_ddq;
            float *Dr = (float *)((char *)idx->ws_ddq + qb);
            int64_t *Ir = (int64_t *)((char *)idx->ws_ddq + qb + db);
            rc = lemon_gather_query_rows(q_dev, rep, U, idx->d, qrep, stream);
            if (rc) return rc;
            const int saved = idx->qdedup;
            idx->qdedup = 0;
            rc = lemon_search_internal(idx, qrep, U, k, Dr, Ir, stream);
            idx->qdedup = saved;
            if (rc) return rc;
            idx->last.nq_distinct = U;
            return lemon_expand_results(Dr, Ir, grp, nq, k, D_dev, I_dev, stream);
        }
    }
    int algo = idx->algo;
    if (k > LEMON_MAX_K) algo = LEMON_ALGO_F32_MFMA;     // deep lists: key-bounded passes of the exact scan (knn_f32.hip)
    if (algo == LEMON_ALGO_AUTO) algo = lemon_auto_choose(idx, q_dev, nq, k, stream);
    int rc = (algo == LEMON_ALGO_BF16_FILTER) ? lemon_search_bf16(idx, q_dev, nq, k, D_dev, I_dev, stream)
                                              : lemon_search_f32(idx, q_dev, nq, k, D_dev, I_dev, stream);
    idx->last.nq_distinct = nq;
    return rc;
}

extern "C" int lemon_index_search(lemon_index_t *idx, const float *q_dev, int64_t nq, int k,
                                  float *D_dev, int64_t *I_dev, void *stream) {
    return lemon_search_internal(idx, q_dev, nq, k, D_dev, I_dev, (hipStream_t)stream);
}

// ---- K4: neighbour finalisation ----------------------------------------------------------
// One thread per (query, slot): self-exclusion offset, sign convention, cross-modal distance by
// the chain contract on gathered rows, dists_tr gather.  run_lemon.py:256-289,303,306.
struct NbParams {
    const float *img_tr, *txt_tr, *dists_tr, *q_img, *q_txt;
    const float *Dn, *Dm;       // [nq, ks] raw search output
    const int64_t *In, *Im;
    const uint8_t *in_db;
    const int32_t *tr_lab, *q_lab;
    float *D_n, *dists_n, *dists_tr_n, *D_m, *dists_m, *dists_tr_m;
    int64_t *I_n, *I_m;
    int64_t nq;
    int d, k, ks, metric, drop_self, discrete;
};

template <bool L2>
__device__ __forceinline__ float chain_dist(const float *__restrict__ a, const float *__restrict__ b, int d) {
    float acc = 0.0f;
    if ((d & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0) {
        const float4 *a4 = reinterpret_cast<const float4 *>(a);
        const float4 *b4 = reinterpret_cast<const float4 *>(b);
        for (int c = 0; c < d / 4; ++c) {
            const float4 u = a4[c], v = b4[c];
            if (L2) {
                float t;
                t = u.x - v.x; acc = __builtin_fmaf(t, t, acc);
                t = u.y - v.y; acc = __builtin_fmaf(t, t, acc);
                t = u.z - v.z; acc = __builtin_fmaf(t, t, acc);
                t = u.w - v.w; acc = __builtin_fmaf(t, t, acc);
            } else {
                acc = __builtin_fmaf(u.x, v.x, acc);
                acc = __builtin_fmaf(u.y, v.y, acc);
                acc = __builtin_fmaf(u.z, v.z, acc);
                acc = __builtin_fmaf(u.w, v.w, acc);
            }
        }
    } else {
        for (int c = 0; c < d; ++c) {
            if (L2) { float t = a[c] - b[c]; acc = __builtin_fmaf(t, t, acc); }
            else acc = __builtin_fmaf(a[c], b[c], acc);
        }
    }
    return L2 ? acc : 1.0f - acc;
}

// Wave-cooperative chained distance.  Lane l owns the row pair (a_l, b_l) and must walk it in ascending
// k with ONE fmaf chain (the numeric contract), but 64 lanes reading 64 different rows 16 B at a time
// touch 64 cache lines per instruction and thrash the 32 KB L1.  So rows are fetched 128 B (one line) at
// a time with full-line loads -- lane (r8, c) fetches chunk c of rows r8, r8+8, ... of the wave -- and
// transposed through LDS (row pitch 36 floats: conflict-free 128-bit reads); the next 128 B are in
// flight while the current ones are consumed.  Needs d % 4 == 0 and 16-byte aligned rows.
constexpr int NB_PITCH = 36;
template <bool L2>
__device__ __forceinline__ float chain_dist_wave(const float *a_row, const float *b_row, int d,
                                                 float *__restrict__ lds /* [2][64][NB_PITCH] */, int lane) {
    const int r8 = lane >> 3, c = lane & 7;
    // named registers, not arrays (hipcc keeps arrays that are filled under a condition in scratch)
#define NB_PTR(i) \
    const float *pa##i = reinterpret_cast<const float *>(__shfl((unsigned long long)(uintptr_t)a_row, r8 + 8 * i)) + 4 * c; \
    const float *pb##i = reinterpret_cast<const float *>(__shfl((unsigned long long)(uintptr_t)b_row, r8 + 8 * i)) + 4 * c;
    NB_PTR(0) NB_PTR(1) NB_PTR(2) NB_PTR(3) NB_PTR(4) NB_PTR(5) NB_PTR(6) NB_PTR(7)
#undef NB_PTR
    float4 va0, va1, va2, va3, va4, va5, va6, va7, vb0, vb1, vb2, vb3, vb4, vb5, vb6, vb7;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#define NB_LOAD(i, koff) \
    va##i = *reinterpret_cast<const float4 *>(pa##i + (koff)); vb##i = *reinterpret_cast<const float4 *>(pb##i + (koff));
#define NB_ZERO(i) va##i = zero; vb##i = zero;
#define NB_FETCH(koff)                                                                              \
    do {                                                                                            \
        if ((koff) + 4 * c < d) {                                                                   \
            NB_LOAD(0, koff) NB_LOAD(1, koff) NB_LOAD(2, koff) NB_LOAD(3, koff)                     \
            NB_LOAD(4, koff) NB_LOAD(5, koff) NB_LOAD(6, koff) NB_LOAD(7, koff)                     \
        } else {                                                                                    \
            NB_ZERO(0) NB_ZERO(1) NB_ZERO(2) NB_ZERO(3) NB_ZERO(4) NB_ZERO(5) NB_ZERO(6) NB_ZERO(7) \
        }                                                                                           \
    } while (0)
#define NB_STORE(i)                                                                    \
    *reinterpret_cast<float4 *>(&la[(r8 + 8 * i) * NB_PITCH + 4 * c]) = va##i;         \
    *reinterpret_cast<float4 *>(&lb[(r8 + 8 * i) * NB_PITCH + 4 * c]) = vb##i;
    NB_FETCH(0);
    float acc = 0.0f;
    float *la = lds, *lb = lds + 64 * NB_PITCH;
    for (int k0 = 0; k0 < d; k0 += 32) {
        NB_STORE(0) NB_STORE(1) NB_STORE(2) NB_STORE(3) NB_STORE(4) NB_STORE(5) NB_STORE(6) NB_STORE(7)
        const int kn = k0 + 32;
        if (kn < d) NB_FETCH(kn);                        // wave-uniform: next 128 B of every row
        __builtin_amdgcn_wave_barrier();
        const int lim = d - k0 < 32 ? d - k0 : 32;       // wave-uniform; multiple of 4
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (4 * u < lim) {
                const float4 x = *reinterpret_cast<const float4 *>(&la[lane * NB_PITCH + 4 * u]);
                const float4 y = *reinterpret_cast<const float4 *>(&lb[lane * NB_PITCH + 4 * u]);
                if (L2) {
                    float t;
                    t = x.x - y.x; acc = __builtin_fmaf(t, t, acc);
                    t = x.y - y.y; acc = __builtin_fmaf(t, t, acc);
                    t = x.z - y.z; acc = __builtin_fmaf(t, t, acc);
                    t = x.w - y.w; acc = __builtin_fmaf(t, t, acc);
                } else {
                    acc = __builtin_fmaf(x.x, y.x, acc);
                    acc = __builtin_fmaf(x.y, y.y, acc);
                    acc = __builtin_fmaf(x.z, y.z, acc);
                    acc = __builtin_fmaf(x.w, y.w, acc);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef NB_STORE
#undef NB_FETCH
#undef NB_ZERO
#undef NB_LOAD
    return L2 ? acc : 1.0f - acc;
}

// STAGED: wave-cooperative row fetches (d % 4 == 0); otherwise every lane walks its rows from global.
template <bool L2, bool STAGED>
__global__ __launch_bounds__(128) void k_neighbors_finalize(NbParams p) {
    __shared__ __attribute__((aligned(16))) float s_rows[STAGED ? 2 : 1][STAGED ? 2 * 64 * NB_PITCH : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = p.nq * p.k;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = t0 < total;
    if (!STAGED && !valid) return;
    const int64_t t = valid ? t0 : total - 1;            // staged: idle lanes shadow the last pair
    const int64_t i = t / p.k;
    const int j = (int)(t % p.k);
    const int off = (p.drop_self && (p.in_db == nullptr || p.in_db[i])) ? 1 : 0;
    const int64_t src = i * p.ks + off + j;
    const int64_t jn = p.In[src], jm = p.Im[src];
    const float dn = p.Dn[src], dm = p.Dm[src];
    const float *vi = p.q_img + i * (int64_t)p.d;
    const float *ti = p.q_txt + i * (int64_t)p.d;
    const float qnan = __builtin_nanf("");
    float dist_n = qnan, dist_m = qnan;
    if (!p.discrete) {                                   // uniform
        const float *yn = jn >= 0 ? p.txt_tr + jn * (int64_t)p.d : ti;
        dist_n = STAGED ? chain_dist_wave<L2>(ti, yn, p.d, s_rows[wave], lane) : chain_dist<L2>(ti, yn, p.d);
    }
    {
        const float *xm = jm >= 0 ? p.img_tr + jm * (int64_t)p.d : vi;
        dist_m = STAGED ? chain_dist_wave<L2>(vi, xm, p.d, s_rows[wave], lane) : chain_dist<L2>(vi, xm, p.d);
    }
    if (!valid) return;
    if (p.I_n) p.I_n[t] = jn;
    if (p.I_m) p.I_m[t] = jm;
    if (jn < 0) { p.D_n[t] = dn; p.dists_n[t] = qnan; p.dists_tr_n[t] = qnan; }
    else {
        if (p.discrete) {
            p.D_n[t] = dn;
            p.dists_n[t] = 1.0f - (float)(p.tr_lab[jn] == p.q_lab[i]);
        } else {
            p.D_n[t] = L2 ? dn : -dn;
            p.dists_n[t] = dist_n;
        }
        p.dists_tr_n[t] = p.dists_tr[jn];
    }
    if (jm < 0) { p.D_m[t] = dm; p.dists_m[t] = qnan; p.dists_tr_m[t] = qnan; }
    else {
        p.D_m[t] = L2 ? dm : -dm;
        p.dists_m[t] = dist_m;
        p.dists_tr_m[t] = p.dists_tr[jm];
    }
}

extern "C" int lemon_neighbors(lemon_index_t *idx_img, lemon_index_t *idx_txt, const float *dists_tr_dev,
                               const float *q_img_dev, const float *q_txt_dev, int64_t nq, int k,
                               int drop_self, const uint8_t *in_db_dev, int discrete,
                               const int32_t *tr_label_id_dev, const int32_t *q_label_id_dev,
                               float *d1_dev, float *D_n_dev, float *dists_n_dev, float *dists_tr_n_dev,
                               int64_t *I_n_dev, float *D_m_dev, float *dists_m_dev, float *dists_tr_m_dev,
                               int64_t *I_m_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LEMON_REQUIRE(idx_img && idx_txt, "index handles");
    LEMON_REQUIRE(idx_img->d == idx_txt->d && idx_img->n == idx_txt->n && idx_img->metric == idx_txt->metric,
                  "image and text index must agree in d, ntotal and metric");
    LEMON_REQUIRE(nq >= 0, "nq >= 0");
    const int ks = k + (drop_self ? 1 : 0);
    LEMON_REQUIRE(k >= 1 && ks <= LEMON_MAX_K, "1 <= k, k + drop_self <= LEMON_MAX_K");
    if (nq == 0) return LEMON_OK;
    LEMON_REQUIRE(dists_tr_dev && q_img_dev && q_txt_dev && d1_dev && D_n_dev && dists_n_dev &&
                      dists_tr_n_dev && D_m_dev && dists_m_dev && dists_tr_m_dev, "null pointer");
    LEMON_REQUIRE(!discrete || (tr_label_id_dev && q_label_id_dev), "label ids required when discrete");
    // raw search outputs live in the image index's neighbour workspace: [2][nq, ks]
    const int64_t need = 2 * nq * ks;
    if (need > idx_img->ws_nb) {
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (idx_img->ws_D) (void)hipFree(idx_img->ws_D);
        if (idx_img->ws_I) (void)hipFree(idx_img->ws_I);
        idx_img->ws_D = nullptr; idx_img->ws_I = nullptr; idx_img->ws_nb = 0;
        if (hipMalloc(&idx_img->ws_D, (size_t)need * sizeof(float)) != hipSuccess ||
            hipMalloc(&idx_img->ws_I, (size_t)need * sizeof(int64_t)) != hipSuccess) {
            lemon_set_error("neighbour workspace allocation failed");
            return LEMON_E_NOMEM;
        }
        idx_img->ws_nb = need;
    }
    float *Dn = idx_img->ws_D, *Dm = idx_img->ws_D + nq * ks;
    int64_t *In = idx_img->ws_I, *Im = idx_img->ws_I + nq * ks;
    // The image-side queries of the LEMoN loop are embeddings of distinct images: looking for duplicates among them costs a
    // row hash, a radix sort and a host synchronisation per call for nothing, so de-duplication is applied to the TEXT side
    // only (class datasets: C distinct prompts among nq queries).  A caller that does have duplicate image
    // queries gets them folded by lemon_index_search on the handle itself.
    const int img_dedup = idx_img->qdedup;
    idx_img->qdedup = 0;
    int rc = lemon_search_internal(idx_img, q_img_dev, nq, ks, Dn, In, stream);   // run_lemon.py:235
    idx_img->qdedup = img_dedup;
    if (rc) return rc;
    rc = lemon_search_internal(idx_txt, q_txt_dev, nq, ks, Dm, Im, stream);       // run_lemon.py:236
    if (rc) return rc;
    rc = lemon_paired_distance(idx_img->metric, q_img_dev, q_txt_dev, nq, idx_img->d, d1_dev, stream);
    if (rc) return rc;
    NbParams p;
    p.img_tr = idx_img->x; p.txt_tr = idx_txt->x; p.dists_tr = dists_tr_dev;
    p.q_img = q_img_dev; p.q_txt = q_txt_dev;
    p.Dn = Dn; p.Dm = Dm; p.In = In; p.Im = Im;
    p.in_db = in_db_dev; p.tr_lab = tr_label_id_dev; p.q_lab = q_label_id_dev;
    p.D_n = D_n_dev; p.dists_n = dists_n_dev; p.dists_tr_n = dists_tr_n_dev;
    p.D_m = D_m_dev; p.dists_m = dists_m_dev; p.dists_tr_m = dists_tr_m_dev;
    p.I_n = I_n_dev; p.I_m = I_m_dev;
    p.nq = nq; p.d = idx_img->d; p.k = k; p.ks = ks; p.metric = idx_img->metric;
    p.drop_self = drop_self; p.discrete = discrete;
    const int64_t total = nq * k;
    {
        const bool staged = (p.d & 3) == 0 &&
                            ((((uintptr_t)p.img_tr) | ((uintptr_t)p.txt_tr) | ((uintptr_t)p.q_img) | ((uintptr_t)p.q_txt)) & 15) == 0;
        const dim3 grid((unsigned)((total + 127) / 128)), block(128);
        const bool l2m = p.metric == LEMON_METRIC_L2;
        if (staged) {
            if (l2m) hipLaunchKernelGGL((k_neighbors_finalize<true, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_neighbors_finalize<false, true>), grid, block, 0, stream, p);
        } else {
            if (l2m) hipLaunchKernelGGL((k_neighbors_finalize<true, false>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_neighbors_finalize<false, false>), grid, block, 0, stream, p);
        }
    }
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

// ---- discrepancy baselines (lib/baselines/discrepancy_baseline.py:164-242) ------------------
// One wavefront per query; lanes stride over the (neighbour, second-order neighbour) or
// (neighbour, neighbour) pairs, each pair is a chain distance on gathered rows; float64 sums.
struct DiscParams {
    const float *E, *qv;
    const int64_t *Im, *Ic;
    float *out;
    int64_t nq;
    int d, k, kq, kc, method;
};

template <bool STAGED>
__global__ __launch_bounds__(128) void k_discrepancy(DiscParams p) {
    __shared__ __attribute__((aligned(16))) float s_rows[STAGED ? 2 : 1][STAGED ? 2 * 64 * NB_PITCH : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 2 + wave;
    if (i >= p.nq) return;                               // wave-uniform
    const int64_t *im = p.Im + i * p.kq;
    double total = 0.0;
    int count = 0;
    const int pairs = p.method == 0 ? p.kq * p.kc : p.kq * p.kq;
    for (int tb = 0; tb < pairs; tb += 64) {
        const int t = tb + lane;
        bool active = t < pairs;
        const float *ra = p.E, *rb = p.E;                // idle lanes read row 0 (always present: n >= 1)
        if (active) {
            const int a = p.method == 0 ? t / p.kc : t / p.kq, b = p.method == 0 ? t % p.kc : t % p.kq;
            const int64_t j = im[a];
            if (j < 0) active = false;
            else if (p.method == 0) {
                const int64_t l = p.Ic[j * p.kc + b];
                if (l < 0 || l == j) active = false;
                else { ra = p.E + l * (int64_t)p.d; rb = p.qv + i * (int64_t)p.d; }
            } else {
                const int64_t l = im[b];
                if (l < 0) active = false;
                else { ra = p.E + j * (int64_t)p.d; rb = p.E + l * (int64_t)p.d; }
            }
        }
        const float dist = STAGED ? chain_dist_wave<false>(ra, rb, p.d, s_rows[wave], lane)
                                  : (active ? chain_dist<false>(ra, rb, p.d) : 0.0f);
        if (active) { total += (double)dist; if (p.method == 0) ++count; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { total += __shfl_xor(total, off); count += __shfl_xor(count, off); }
    if (lane == 0)
        p.out[i] = p.method == 0 ? (float)(count ? total / (double)count : (double)__builtin_nanf(""))
                                 : (float)(total / ((double)p.k * (double)p.k));
}

extern "C" int lemon_discrepancy(int method, lemon_index_t *idx_txt, const float *E_tr_dev, const float *qv_dev,
                                 const float *q_txt_dev, int64_t nq, int k, int is_train, float *out_dev,
                                 void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LEMON_REQUIRE(method == 0 || method == 1, "method: 0 dis, 1 div");
    LEMON_REQUIRE(idx_txt && idx_txt->metric == LEMON_METRIC_IP, "text index must be an IndexFlatIP");
    LEMON_REQUIRE(nq >= 0 && k >= 1 && k + 1 <= LEMON_MAX_K, "nq >= 0, 1 <= k < LEMON_MAX_K");
    if (nq == 0) return LEMON_OK;
    LEMON_REQUIRE(E_tr_dev && q_txt_dev && out_dev && (method == 1 || qv_dev), "null pointer");
    const int kc = k + 1, kq = k + (is_train ? 1 : 0);
    const int64_t n = idx_txt->n;
    // scratch: cache [n, kc] (dis only) + Im [nq, kq] (+ the D arrays the searches return)
    const int64_t need = (method == 0 ? n * kc : 0) + nq * kq;
    if (need > idx_txt->ws_nb) {
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (idx_txt->ws_D) (void)hipFree(idx_txt->ws_D);
        if (idx_txt->ws_I) (void)hipFree(idx_txt->ws_I);
        idx_txt->ws_D = nullptr; idx_txt->ws_I = nullptr; idx_txt->ws_nb = 0;
        if (hipMalloc(&idx_txt->ws_D, (size_t)need * sizeof(float)) != hipSuccess ||
            hipMalloc(&idx_txt->ws_I, (size_t)need * sizeof(int64_t)) != hipSuccess) {
            lemon_set_error("discrepancy workspace allocation failed");
            return LEMON_E_NOMEM;
        }
        idx_txt->ws_nb = need;
    }
    int64_t *Ic = idx_txt->ws_I, *Im = idx_txt->ws_I + (method == 0 ? n * kc : 0);
    float *Dc = idx_txt->ws_D, *Dm = idx_txt->ws_D + (method == 0 ? n * kc : 0);
    int rc;
    if (method == 0) {   // cache of the DB's own text neighbours, discrepancy_baseline.py:165-169
        rc = lemon_search_internal(idx_txt, idx_txt->x, n, kc, Dc, Ic, stream);
        if (rc) return rc;
    }
    rc = lemon_search_internal(idx_txt, q_txt_dev, nq, kq, Dm, Im, stream);   // :209
    if (rc) return rc;
    DiscParams p;
    p.E = E_tr_dev; p.qv = qv_dev; p.Im = Im; p.Ic = Ic; p.out = out_dev; p.nq = nq;
    p.d = idx_txt->d; p.k = k; p.kq = kq; p.kc = kc; p.method = method;
    {
        const bool staged = (p.d & 3) == 0 && ((((uintptr_t)p.E) | ((uintptr_t)p.qv)) & 15) == 0;
        const dim3 grid((unsigned)((nq + 1) / 2)), block(128);
        if (staged) hipLaunchKernelGGL(k_discrepancy<true>, grid, block, 0, stream, p);
        else hipLaunchKernelGGL(k_discrepancy<false>, grid, block, 0, stream, p);
    }
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
