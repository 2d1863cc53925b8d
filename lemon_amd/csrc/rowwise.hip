// rowwise.hip -- HBM-bound row-wise kernels of the LEMoN hot path (gfx950).
//   K1 l2_normalize_rows   lib/utils/utils.py:39-40
//   K2 paired_distance     run_lemon.py:169,173,250-253
//   K2' d1_normalized      run_lemon.py:244-248
//   K5 lemon_score         lib/metrics/utils.py:47-82
// All are bound by HBM (2*N*d*4 B for K1, 2*N*d*4 B read for K2, 6*N*k*4 B for K5).
#include "common.hpp"

// ---------------------------------------------------------------------------------
// K1: one wavefront per row, float4 coalesced reads, float64 sum of squares.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_normalize_rows(const float *__restrict__ x, int64_t n, int d,
                                                        float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *r = x + row * (int64_t)d;
    float *o = y + row * (int64_t)d;
    double ss = 0.0;
    const bool vec = ((d & 3) == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)y) & 15) == 0);
    if (vec) {
        const float4 *r4 = reinterpret_cast<const float4 *>(r);
        for (int c = lane; c < d / 4; c += 64) {
            float4 v = r4[c];
            ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int c = lane; c < d; c += 64) { float v = r[c]; ss += (double)v * v; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float nrm = (float)sqrt(ss);
    const float den = nrm > 1e-12f ? nrm : 1e-12f;
    if (vec) {
        const float4 *r4 = reinterpret_cast<const float4 *>(r);
        float4 *o4 = reinterpret_cast<float4 *>(o);
        for (int c = lane; c < d / 4; c += 64) {
            float4 v = r4[c];
            v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
            o4[c] = v;
        }
    } else {
        for (int c = lane; c < d; c += 64) o[c] = r[c] / den;
    }
}

// ---------------------------------------------------------------------------------
// chain row-dot through an LDS transpose: a block stages a [64 rows x 64 cols] slab of a and b
// with coalesced loads, then lane r walks row r in ascending k (the fmaf chain of the numeric
// contract).  MODE 0: dot(a,b)  MODE 1: 1 - dot(a,b)  MODE 2: sum (a-b)^2  MODE 3: sqrt(sum (a-b)^2)
// MODE 4: sum |a-b|  MODE 5: 1 - dot(a,b)/(|a||b|)   (3-5: DistanceEvaluator.our_metric,
// lib/metrics/distance_metrics.py:48-73, which takes the diagonal of a full pairwise matrix)
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void k_rowchain(const float *__restrict__ a, const float *__restrict__ b,
                                                 int64_t n, int d, float *__restrict__ out) {
    __shared__ float sa[64][65];
    __shared__ float sb[64][65];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    float acc = 0.0f, na = 0.0f, nb = 0.0f;
    for (int k0 = 0; k0 < d; k0 += 64) {
        // coalesced: for each of the 64 rows, 64 lanes read 64 consecutive floats
        for (int r = 0; r < 64; ++r) {
            int64_t row = row0 + r;
            int k = k0 + lane;
            float va = 0.0f, vb = 0.0f;
            if (row < n && k < d) { va = a[row * (int64_t)d + k]; vb = b[row * (int64_t)d + k]; }
            sa[r][lane] = va; sb[r][lane] = vb;
        }
        __syncthreads();
        const int kmax = (d - k0) < 64 ? (d - k0) : 64;
        for (int k = 0; k < kmax; ++k) {
            float va = sa[lane][k], vb = sb[lane][k];
            if (MODE == 2 || MODE == 3) { float t = va - vb; acc = __builtin_fmaf(t, t, acc); }
            else if (MODE == 4) acc += fabsf(va - vb);
            else acc = __builtin_fmaf(va, vb, acc);
            if (MODE == 5) { na = __builtin_fmaf(va, va, na); nb = __builtin_fmaf(vb, vb, nb); }
        }
        __syncthreads();
    }
    int64_t row = row0 + lane;
    if (row < n) {
        float r = acc;
        if (MODE == 1) r = 1.0f - acc;
        if (MODE == 3) r = sqrtf(acc);
        if (MODE == 5) r = 1.0f - acc / fmaxf(sqrtf(na) * sqrtf(nb), 1e-30f);
        out[row] = r;
    }
}

int lemon_rowdot_chain(const float *a, const float *b, int64_t n, int d, float *out, hipStream_t s) {
    if (n <= 0) return LEMON_OK;
    hipLaunchKernelGGL(k_rowchain<0>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a, b, n, d, out);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

// ---------------------------------------------------------------------------------
// K2': --normalize_d1.  One wavefront per query; lane c walks class prompt c (chain), then a
// wave softmax in float32 (scipy.special.softmax on float32 input).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_d1_normalized(int metric, const float *__restrict__ q, int64_t n, int d,
                                                      const float *__restrict__ cls, int C,
                                                      const int32_t *__restrict__ lab, float *__restrict__ d1) {
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const float *v = q + i * (int64_t)d;
    const int mylab = lab[i];
    float mx = -FLT_MAX, mine = 0.0f;
    // pass 1: distances for classes lane, lane+64, ...; keep in a small register array (C <= 64*16)
    float z[16];
    int nz = 0;
    for (int c = lane; c < C && nz < 16; c += 64, ++nz) {
        const float *t = cls + (int64_t)c * d;
        float acc = 0.0f;
        if (metric == LEMON_METRIC_IP) {
            for (int k = 0; k < d; ++k) acc = __builtin_fmaf(v[k], t[k], acc);
            acc = 1.0f - acc;
        } else {
            for (int k = 0; k < d; ++k) { float u = v[k] - t[k]; acc = __builtin_fmaf(u, u, acc); }
        }
        z[nz] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float s = 0.0f;
    int j = 0;
    for (int c = lane; c < C && j < 16; c += 64, ++j) {
        float e = expf(z[j] - mx);
        s += e;
        if (c == mylab) mine = e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); mine += __shfl_xor(mine, off); }
    if (lane == 0) d1[i] = mine / s;
}

// ---------------------------------------------------------------------------------
// Zero-shot "CLIP logits" confidence (lib/baselines/train_zero_shot_clip_baseline.py:207-224): for one image, the
// distance of its UN-normalised embedding to every class-prompt embedding by DistanceEvaluator.our_metric
// (lib/metrics/distance_metrics.py:48-73: 1 - cosine similarity | euclidean (not squared) | manhattan),
// conf = softmax_c(1 - dist_c)[noisy label].  One wavefront per image; lane c walks class prompt c; float32 softmax.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_class_confidence(int kind, const float *__restrict__ img, int64_t n, int d,
                                                         const float *__restrict__ cls, int C,
                                                         const int32_t *__restrict__ lab, float *__restrict__ conf) {
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const float *v = img + i * (int64_t)d;
    const int mylab = lab[i];
    float vv = 0.0f;
    if (kind == 0) for (int k = 0; k < d; ++k) vv = __builtin_fmaf(v[k], v[k], vv);
    float z[16];
    float mx = -FLT_MAX, mine = 0.0f;
    int nz = 0;
    for (int c = lane; c < C && nz < 16; c += 64, ++nz) {
        const float *t = cls + (int64_t)c * d;
        float dist;
        if (kind == 0) {
            float dot = 0.0f, tt = 0.0f;
            for (int k = 0; k < d; ++k) { dot = __builtin_fmaf(v[k], t[k], dot); tt = __builtin_fmaf(t[k], t[k], tt); }
            dist = 1.0f - dot / (sqrtf(vv) * sqrtf(tt));
        } else if (kind == 1) {
            float acc = 0.0f;
            for (int k = 0; k < d; ++k) { const float u = v[k] - t[k]; acc = __builtin_fmaf(u, u, acc); }
            dist = sqrtf(acc);
        } else {
            float acc = 0.0f;
            for (int k = 0; k < d; ++k) acc += fabsf(v[k] - t[k]);
            dist = acc;
        }
        z[nz] = 1.0f - dist;
        mx = fmaxf(mx, z[nz]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float s = 0.0f;
    int j = 0;
    for (int c = lane; c < C && j < 16; c += 64, ++j) {
        const float e = expf(z[j] - mx);
        s += e;
        if (c == mylab) mine = e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); mine += __shfl_xor(mine, off); }
    if (lane == 0) conf[i] = mine / s;
}

// ---------------------------------------------------------------------------------
// K5: score aggregation, one thread per sample, float64 like the oracle.
// ---------------------------------------------------------------------------------
struct ScoreHP { double beta, gamma, t1n, t2n, t1m, t2m; };

__global__ __launch_bounds__(256) void k_score(const float *__restrict__ d1, const float *__restrict__ Dn,
                                               const float *__restrict__ trn, const float *__restrict__ dn,
                                               const float *__restrict__ Dm, const float *__restrict__ trm,
                                               const float *__restrict__ dm, int64_t n, int k, ScoreHP hp,
                                               double *__restrict__ score, double *__restrict__ o_dn,
                                               double *__restrict__ o_dm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double sn = 0.0, sm = 0.0;
    for (int j = 0; j < k; ++j) {
        const int64_t o = i * (int64_t)k + j;
        sn += exp(-hp.t1n * (double)Dn[o]) * exp(-hp.t2n * (double)trn[o]) * (double)dn[o];
        sm += exp(-hp.t1m * (double)Dm[o]) * exp(-hp.t2m * (double)trm[o]) * (double)dm[o];
    }
    const double a = sn / (double)k, b = sm / (double)k;
    if (o_dn) o_dn[i] = a;
    if (o_dm) o_dm[i] = b;
    score[i] = (double)d1[i] + hp.beta * a + hp.gamma * b;
}


// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
extern "C" int lemon_normalize_rows(const float *x_dev, int64_t n, int d, float *y_dev, void *stream) {
    LEMON_REQUIRE(n >= 0 && d > 0, "n >= 0 and d > 0");
    if (n == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev && y_dev, "null pointer");
    hipLaunchKernelGGL(k_normalize_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x_dev, n, d, y_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_paired_distance(int metric, const float *a_dev, const float *b_dev, int64_t n, int d,
                                     float *out_dev, void *stream) {
    LEMON_REQUIRE(metric == LEMON_METRIC_IP || metric == LEMON_METRIC_L2, "metric");
    LEMON_REQUIRE(n >= 0 && d > 0, "n >= 0 and d > 0");
    if (n == 0) return LEMON_OK;
    LEMON_REQUIRE(a_dev && b_dev && out_dev, "null pointer");
    dim3 grid((unsigned)((n + 63) / 64));
    if (metric == LEMON_METRIC_IP)
        hipLaunchKernelGGL(k_rowchain<1>, grid, dim3(64), 0, (hipStream_t)stream, a_dev, b_dev, n, d, out_dev);
    else
        hipLaunchKernelGGL(k_rowchain<2>, grid, dim3(64), 0, (hipStream_t)stream, a_dev, b_dev, n, d, out_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_paired_metric(int kind, const float *a_dev, const float *b_dev, int64_t n, int d,
                                   float *out_dev, void *stream) {
    LEMON_REQUIRE(kind >= 0 && kind <= 2, "kind: 0 cosine, 1 euclidean, 2 manhattan");
    LEMON_REQUIRE(n >= 0 && d > 0, "n >= 0 and d > 0");
    if (n == 0) return LEMON_OK;
    LEMON_REQUIRE(a_dev && b_dev && out_dev, "null pointer");
    dim3 grid((unsigned)((n + 63) / 64));
    if (kind == 0) hipLaunchKernelGGL(k_rowchain<5>, grid, dim3(64), 0, (hipStream_t)stream, a_dev, b_dev, n, d, out_dev);
    else if (kind == 1) hipLaunchKernelGGL(k_rowchain<3>, grid, dim3(64), 0, (hipStream_t)stream, a_dev, b_dev, n, d, out_dev);
    else hipLaunchKernelGGL(k_rowchain<4>, grid, dim3(64), 0, (hipStream_t)stream, a_dev, b_dev, n, d, out_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_d1_normalized(int metric, const float *q_img_dev, int64_t n, int d,
                                   const float *cls_txt_dev, int C, const int32_t *noisy_label_dev,
                                   float *d1_dev, void *stream) {
    LEMON_REQUIRE(metric == LEMON_METRIC_IP || metric == LEMON_METRIC_L2, "metric");
    LEMON_REQUIRE(n >= 0 && d > 0 && C > 0 && C <= 1024, "n >= 0, d > 0, 0 < C <= 1024");
    if (n == 0) return LEMON_OK;
    LEMON_REQUIRE(q_img_dev && cls_txt_dev && noisy_label_dev && d1_dev, "null pointer");
    hipLaunchKernelGGL(k_d1_normalized, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, metric,
                       q_img_dev, n, d, cls_txt_dev, C, noisy_label_dev, d1_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_class_confidence(int kind, const float *img_dev, int64_t n, int d, const float *cls_txt_dev, int C,
                                      const int32_t *noisy_label_dev, float *conf_dev, void *stream) {
    LEMON_REQUIRE(kind >= 0 && kind <= 2, "kind: 0 cosine, 1 euclidean, 2 manhattan");
    LEMON_REQUIRE(n >= 0 && d > 0 && C > 0 && C <= 1024, "n >= 0, d > 0, 0 < C <= 1024");
    if (n == 0) return LEMON_OK;
    LEMON_REQUIRE(img_dev && cls_txt_dev && noisy_label_dev && conf_dev, "null pointer");
    hipLaunchKernelGGL(k_class_confidence, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, kind, img_dev, n, d,
                       cls_txt_dev, C, noisy_label_dev, conf_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_score(const float *d1_dev, const float *D_n_dev, const float *dists_tr_n_dev,
                           const float *dists_n_dev, const float *D_m_dev, const float *dists_tr_m_dev,
                           const float *dists_m_dev, int64_t n, int k, const double hp[6],
                           double *score_dev, double *d_n_dev, double *d_m_dev, void *stream) {
    LEMON_REQUIRE(n >= 0 && k > 0, "n >= 0 and k > 0");
    if (n == 0) return LEMON_OK;
    LEMON_REQUIRE(d1_dev && D_n_dev && dists_tr_n_dev && dists_n_dev && D_m_dev && dists_tr_m_dev &&
                      dists_m_dev && hp && score_dev, "null pointer");
    ScoreHP h = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
    hipLaunchKernelGGL(k_score, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d1_dev, D_n_dev, dists_tr_n_dev, dists_n_dev, D_m_dev, dists_tr_m_dev, dists_m_dev,
                       n, k, h, score_dev, d_n_dev, d_m_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
