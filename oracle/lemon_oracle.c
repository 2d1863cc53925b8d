/*
 * lemon_oracle.c -- CPU restatement of LEMoN's embed->kNN->score hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (lemon_amd/, include/,
 * the C-ABI library) may import, link or execute this file.  It is used by
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg as the
 * CHECKER for the HIP path, never as the thing measured as `value` or shipped.
 *
 * PARITY STATUS
 *   - score aggregation, paired distances, normalisation, noise/splits: pinned
 *     by golden vectors generated from the importable reference modules
 *     (tests/golden/, tools/make_golden.py).
 *   - the per-sample loop (lo_neighbors: self-exclusion, D_n / D_m sign quirk, discrete text metric,
 *     --normalize_d1, DB subset with mixed in_db, record order) and the discrepancy baselines
 *     (lo_discrepancy): pinned by fixtures produced by EXECUTING the reference's own scripts
 *     (/root/reference/run_lemon.py and lib/baselines/discrepancy_baseline.py under runpy, with stand-ins for
 *     faiss / the CLIP weights / the datasets: tools/make_golden_loop.py -> tests/golden/loop_*.npz,
 *     disc_*.npz; tests/test_loop_golden.py, tests/test_disc_golden.py).  The faiss stand-in implements
 *     THIS file's documented contract in numpy, so those fixtures pin the loop around the search, not:
 *   - kNN arithmetic (run_lemon.py:166-176,235-236): the reference delegates to
 *     the third-party package `faiss-gpu` (requirements.txt:21, version NOT
 *     pinned, source not under /root/reference, not installed here) and holds
 *     no test vectors for it => "parity unpinned" at that boundary.  The
 *     restatement below follows faiss's published IndexFlat semantics (exact
 *     search, best-first, int64 labels, squared L2 via the norm expansion,
 *     -1 / +-FLT_MAX padding) with a fully specified numeric contract:
 *
 * NUMERIC CONTRACT ("chain" numerics, shared bit-for-bit with the HIP kernels)
 *   dot(a,b)    : acc = +0.0f; for k = 0..d-1 ascending: acc = fmaf(a[k], b[k], acc)
 *                 (this is exactly what a chain of v_mfma_f32_32x32x2_f32
 *                 instructions computes on gfx950)
 *   sqdiff(a,b) : acc = +0.0f; for k ascending: t = a[k]-b[k]; acc = fmaf(t,t,acc)
 *   IP search   : D = dot(q,x);                      order (D desc, index asc)
 *   L2 search   : D = max(0, fmaf(-2, dot(q,x), dot(q,q)+dot(x,x)));
 *                                                    order (D asc,  index asc)
 *   normalise   : ss = sum_k (double)x[k]^2 (ascending k); y = x / max((float)sqrt(ss), 1e-12f)
 *   ties are ALWAYS broken towards the lower database index.
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define LO_SIMD 1
#else
#define LO_SIMD 0
#endif

#define LO_METRIC_IP 0      /* faiss.IndexFlatIP  / --dist_type cosine    */
#define LO_METRIC_L2 1      /* faiss.IndexFlatL2  / --dist_type euclidean */

/* ------------------------------------------------------------------ */
/* primitives                                                          */
/* ------------------------------------------------------------------ */
static inline float lo_dot(const float *a, const float *b, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}
static inline float lo_sqdiff(const float *a, const float *b, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) { float t = a[k] - b[k]; acc = fmaf(t, t, acc); }
    return acc;
}
float lo_dot_chain(const float *a, const float *b, int d) { return lo_dot(a, b, d); }

/* lib/utils/utils.py:39-40  normalize_vectors = F.normalize(p=2, dim=1), eps 1e-12 */
void lo_normalize_rows(const float *x, int64_t n, int d, float *y) {
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *r = x + i * (int64_t)d;
        double ss = 0.0;
        for (int k = 0; k < d; ++k) ss += (double)r[k] * (double)r[k];
        float nrm = (float)sqrt(ss);
        float den = nrm > 1e-12f ? nrm : 1e-12f;
        float *o = y + i * (int64_t)d;
        for (int k = 0; k < d; ++k) o[k] = r[k] / den;
    }
}

/* run_lemon.py:169 (cosine: 1 - <t,v>), :173 (euclidean: sum (t-v)^2), :250-253 (d_1) */
void lo_paired_distance(int metric, const float *a, const float *b, int64_t n, int d, float *out) {
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *p = a + i * (int64_t)d, *q = b + i * (int64_t)d;
        out[i] = (metric == LO_METRIC_IP) ? 1.0f - lo_dot(p, q, d) : lo_sqdiff(p, q, d);
    }
}

/* ------------------------------------------------------------------ */
/* exact flat search (stand-in for faiss IndexFlat{IP,L2}.search,      */
/* call sites run_lemon.py:167-176,235-236)                            */
/* ------------------------------------------------------------------ */
typedef struct { float v; int64_t i; } lo_ent;

static inline int lo_better(int metric, float av, int64_t ai, float bv, int64_t bi) {
    if (metric == LO_METRIC_IP) { if (av > bv) return 1; if (av < bv) return 0; }
    else                        { if (av < bv) return 1; if (av > bv) return 0; }
    return ai < bi;
}

/* sorted (best first) insertion list of capacity k */
static inline void lo_push(int metric, lo_ent *h, int *cnt, int k, float v, int64_t idx) {
    if (v != v) return;                                  /* NaN never selected */
    if (*cnt == k && !lo_better(metric, v, idx, h[k - 1].v, h[k - 1].i)) return;
    int p = (*cnt < k) ? (*cnt)++ : k - 1;
    while (p > 0 && lo_better(metric, v, idx, h[p - 1].v, h[p - 1].i)) { h[p] = h[p - 1]; --p; }
    h[p].v = v; h[p].i = idx;
}

static inline float lo_l2_from_ip(float ip, float qn, float xn) {
    float v = fmaf(-2.0f, ip, qn + xn);
    return v > 0.0f ? v : 0.0f;
}

/*
 * X [n,d] database, Q [nq,d] queries (row-major f32).  D [nq,k] f32, I [nq,k] i64,
 * best first.  Slots beyond n are padded with I=-1 and D=-FLT_MAX (IP) / +FLT_MAX (L2).
 */
void lo_knn(int metric, const float *X, int64_t n, int d,
            const float *Q, int64_t nq, int k, float *D, int64_t *I) {
    if (k <= 0 || nq <= 0) return;
    const int64_t nb = (n + 7) / 8;
    float *XT = NULL, *xn = NULL;
    /* transposed 8-row blocks: XT[b][kk][lane] so that 8 independent chains run in one vector */
    if (n > 0) {
        /* (C11: the size passed to aligned_alloc must be a multiple of the alignment -- found by the ASan build, tests/test_sanitizers.py) */
        XT = (float *)aligned_alloc(64, (((size_t)nb * d * 8 * sizeof(float)) + 63) / 64 * 64);
        xn = (float *)malloc((size_t)nb * 8 * sizeof(float));
        #pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < nb; ++b)
            for (int l = 0; l < 8; ++l) {
                int64_t r = b * 8 + l;
                for (int kk = 0; kk < d; ++kk)
                    XT[((size_t)b * d + kk) * 8 + l] = (r < n) ? X[r * (int64_t)d + kk] : 0.0f;
                xn[b * 8 + l] = (r < n) ? lo_dot(X + r * (int64_t)d, X + r * (int64_t)d, d) : 0.0f;
            }
    }
    #pragma omp parallel
    {
        lo_ent *heap = (lo_ent *)malloc(sizeof(lo_ent) * (size_t)k * 4);
        #pragma omp for schedule(dynamic, 1)
        for (int64_t q0 = 0; q0 < nq; q0 += 4) {
            int nqt = (int)((nq - q0) < 4 ? (nq - q0) : 4);
            int cnt[4] = {0, 0, 0, 0};
            const float *qp[4]; float qn[4];
            for (int t = 0; t < 4; ++t) {
                qp[t] = Q + (q0 + (t < nqt ? t : 0)) * (int64_t)d;
                qn[t] = lo_dot(qp[t], qp[t], d);
            }
            for (int64_t b = 0; b < nb; ++b) {
                float vals[4][8];
                const float *xt = XT + (size_t)b * d * 8;
#if LO_SIMD
                __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
                for (int kk = 0; kk < d; ++kk) {
                    __m256 xv = _mm256_load_ps(xt + (size_t)kk * 8);
                    a0 = _mm256_fmadd_ps(_mm256_broadcast_ss(qp[0] + kk), xv, a0);
                    a1 = _mm256_fmadd_ps(_mm256_broadcast_ss(qp[1] + kk), xv, a1);
                    a2 = _mm256_fmadd_ps(_mm256_broadcast_ss(qp[2] + kk), xv, a2);
                    a3 = _mm256_fmadd_ps(_mm256_broadcast_ss(qp[3] + kk), xv, a3);
                }
                _mm256_storeu_ps(vals[0], a0); _mm256_storeu_ps(vals[1], a1);
                _mm256_storeu_ps(vals[2], a2); _mm256_storeu_ps(vals[3], a3);
#else
                for (int t = 0; t < 4; ++t)
                    for (int l = 0; l < 8; ++l) {
                        float acc = 0.0f;
                        for (int kk = 0; kk < d; ++kk) acc = fmaf(qp[t][kk], xt[(size_t)kk * 8 + l], acc);
                        vals[t][l] = acc;
                    }
#endif
                for (int t = 0; t < nqt; ++t)
                    for (int l = 0; l < 8; ++l) {
                        int64_t r = b * 8 + l;
                        if (r >= n) break;
                        float v = vals[t][l];
                        if (metric == LO_METRIC_L2) v = lo_l2_from_ip(v, qn[t], xn[r]);
                        lo_push(metric, heap + (size_t)t * k, &cnt[t], k, v, r);
                    }
            }
            for (int t = 0; t < nqt; ++t)
                for (int j = 0; j < k; ++j) {
                    int64_t o = (q0 + t) * (int64_t)k + j;
                    if (j < cnt[t]) { D[o] = heap[(size_t)t * k + j].v; I[o] = heap[(size_t)t * k + j].i; }
                    else { D[o] = (metric == LO_METRIC_IP) ? -FLT_MAX : FLT_MAX; I[o] = -1; }
                }
        }
        free(heap);
    }
    free(XT); free(xn);
}

/* ------------------------------------------------------------------ */
/* per-sample multimodal-neighbour quantities, run_lemon.py:238-307    */
/* ------------------------------------------------------------------ */
/*
 * One call = one split (sname) of the reference's scoring loop, all batches.
 *   img_tr, txt_tr [n_tr,d]  normalised DB embeddings   (run_lemon.py:163-164)
 *   dists_tr [n_tr]                                      (:169 / :173)
 *   q_img, q_txt [nq,d]      normalised query embeddings (:230-233)
 *   drop_self: 1 for the train split (search k+1, :235-236), else 0
 *   in_db [nq] (u8, only read when drop_self): sample_idx in train_indices_in_compr (:258,:278)
 *   discrete: --use_discrete_for_text (:266-267); tr_label_id [n_tr], q_label_id [nq] are
 *             integer ids of the prompt strings compared there.
 * Outputs (each [nq,k] unless noted): d1 [nq], D_n, dists_n, dists_tr_n, I_n, D_m, dists_m,
 * dists_tr_m, I_m with the sign convention of :269-270,285-286 (A14).
 * Neighbours that do not exist (index -1, when n_tr < k+drop_self) yield NaN distances.
 */
void lo_neighbors(int metric, const float *img_tr, const float *txt_tr, const float *dists_tr,
                  int64_t n_tr, int d, const float *q_img, const float *q_txt, int64_t nq, int k,
                  int drop_self, const uint8_t *in_db, int discrete,
                  const int32_t *tr_label_id, const int32_t *q_label_id,
                  float *d1, float *D_n, float *dists_n, float *dists_tr_n, int64_t *I_n,
                  float *D_m, float *dists_m, float *dists_tr_m, int64_t *I_m) {
    const int ks = k + (drop_self ? 1 : 0);
    float *Dn = (float *)malloc(sizeof(float) * (size_t)nq * ks);
    float *Dm = (float *)malloc(sizeof(float) * (size_t)nq * ks);
    int64_t *In = (int64_t *)malloc(sizeof(int64_t) * (size_t)nq * ks);
    int64_t *Im = (int64_t *)malloc(sizeof(int64_t) * (size_t)nq * ks);
    lo_knn(metric, img_tr, n_tr, d, q_img, nq, ks, Dn, In);      /* index_img.search, :235 */
    lo_knn(metric, txt_tr, n_tr, d, q_txt, nq, ks, Dm, Im);      /* index_txt.search, :236 */
    lo_paired_distance(metric, q_img, q_txt, nq, d, d1);         /* d_1, :250-253 */
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i) {
        /* :257-263 / :277-283 -- drop result[0] if the sample is in the DB else drop result[-1] */
        int off = (drop_self && in_db[i]) ? 1 : 0;
        const float *vi = q_img + i * (int64_t)d, *ti = q_txt + i * (int64_t)d;
        for (int j = 0; j < k; ++j) {
            int64_t o = i * (int64_t)k + j;
            int64_t jn = In[i * (int64_t)ks + off + j], jm = Im[i * (int64_t)ks + off + j];
            float dn = Dn[i * (int64_t)ks + off + j], dm = Dm[i * (int64_t)ks + off + j];
            I_n[o] = jn; I_m[o] = jm;
            /* image-neighbour side, :264-273 */
            if (jn < 0) { D_n[o] = dn; dists_n[o] = NAN; dists_tr_n[o] = NAN; }
            else {
                const float *yn = txt_tr + jn * (int64_t)d;
                if (discrete) {                                             /* :266-267 */
                    D_n[o] = dn;                                            /* no negation (A14) */
                    dists_n[o] = 1.0f - (float)(tr_label_id[jn] == q_label_id[i]);
                } else if (metric == LO_METRIC_IP) {                        /* :269-271 */
                    D_n[o] = -dn;
                    dists_n[o] = 1.0f - lo_dot(ti, yn, d);
                } else {                                                    /* :272-273 */
                    D_n[o] = dn;
                    dists_n[o] = lo_sqdiff(ti, yn, d);
                }
                dists_tr_n[o] = dists_tr[jn];                               /* :303 */
            }
            /* text-neighbour side, :275-289 */
            if (jm < 0) { D_m[o] = dm; dists_m[o] = NAN; dists_tr_m[o] = NAN; }
            else {
                const float *xm = img_tr + jm * (int64_t)d;
                if (metric == LO_METRIC_IP) { D_m[o] = -dm; dists_m[o] = 1.0f - lo_dot(vi, xm, d); }
                else                        { D_m[o] = dm;  dists_m[o] = lo_sqdiff(vi, xm, d); }
                dists_tr_m[o] = dists_tr[jm];                               /* :306 */
            }
        }
    }
    free(Dn); free(Dm); free(In); free(Im);
}

/* --normalize_d1, run_lemon.py:244-248: softmax over class prompts of the per-class distance,
 * picked at the noisy label.  cls_txt [C,d] normalised class-prompt embeddings (:180-190). */
void lo_d1_normalized(int metric, const float *q_img, int64_t nq, int d, const float *cls_txt, int C,
                      const int32_t *noisy_label, float *d1) {
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i) {
        const float *v = q_img + i * (int64_t)d;
        float mx = -FLT_MAX;
        float *z = (float *)malloc(sizeof(float) * (size_t)C);
        for (int c = 0; c < C; ++c) {
            const float *t = cls_txt + (int64_t)c * d;
            z[c] = (metric == LO_METRIC_IP) ? 1.0f - lo_dot(v, t, d) : lo_sqdiff(v, t, d);
            if (z[c] > mx) mx = z[c];
        }
        float s = 0.0f;
        for (int c = 0; c < C; ++c) { z[c] = expf(z[c] - mx); s += z[c]; }
        d1[i] = z[noisy_label[i]] / s;
        free(z);
    }
}

/* ------------------------------------------------------------------ */
/* score aggregation, lib/metrics/utils.py:47-82 (vectorised) == :21-45 (loop) */
/* ------------------------------------------------------------------ */
/* hp = {beta, gamma, tau_1_n, tau_2_n, tau_1_m, tau_2_m}.  float64 like NumPy-2 promotion
 * of np.float64 hparams; float32 inputs. */
void lo_score(const float *d1, const float *D_n, const float *dists_tr_n, const float *dists_n,
              const float *D_m, const float *dists_tr_m, const float *dists_m,
              int64_t n, int k, const double *hp,
              double *score, double *d_n_out, double *d_m_out) {
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double sn = 0.0, sm = 0.0;
        for (int j = 0; j < k; ++j) {
            int64_t o = i * (int64_t)k + j;
            sn += exp(-hp[2] * (double)D_n[o]) * exp(-hp[3] * (double)dists_tr_n[o]) * (double)dists_n[o];
            sm += exp(-hp[4] * (double)D_m[o]) * exp(-hp[5] * (double)dists_tr_m[o]) * (double)dists_m[o];
        }
        double dn = sn / (double)k, dm = sm / (double)k;
        if (d_n_out) d_n_out[i] = dn;
        if (d_m_out) d_m_out[i] = dm;
        score[i] = (double)d1[i] + hp[0] * dn + hp[1] * dm;
    }
}

/* ------------------------------------------------------------------ */
/* discrepancy baselines, lib/baselines/discrepancy_baseline.py:164-242 */
/* ------------------------------------------------------------------ */
/*
 * method 0 = dis_* (second-order neighbours through the DB self-kNN cache, :165-169,214-221),
 * method 1 = div_* (mean pairwise distance inside the neighbour set, :222-229; divides by k^2 even
 * when k+1 neighbours were searched on the train split).  E_tr / qv are the embeddings of the
 * modality being scored (x: image, y: text); neighbours always come from the TEXT index (:209).
 * The reference does NOT drop the query itself from I_m on the train split.
 */
void lo_discrepancy(int method, const float *E_tr, const float *txt_tr, int64_t n_tr, int d,
                    const float *qv, const float *q_txt, int64_t nq, int k, int is_train, float *out) {
    const int kc = k + 1, kq = k + (is_train ? 1 : 0);
    float *Dc = (float *)malloc(sizeof(float) * (size_t)n_tr * kc);
    int64_t *Ic = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_tr * kc);
    float *Dm = (float *)malloc(sizeof(float) * (size_t)nq * kq);
    int64_t *Im = (int64_t *)malloc(sizeof(int64_t) * (size_t)nq * kq);
    if (method == 0) lo_knn(LO_METRIC_IP, txt_tr, n_tr, d, txt_tr, n_tr, kc, Dc, Ic);   /* cache, :166 */
    lo_knn(LO_METRIC_IP, txt_tr, n_tr, d, q_txt, nq, kq, Dm, Im);                       /* :209 */
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i) {
        double total = 0.0;
        int64_t count = 0;
        for (int a = 0; a < kq; ++a) {
            const int64_t j = Im[i * kq + a];
            if (j < 0) continue;
            if (method == 0) {
                for (int c = 0; c < kc; ++c) {
                    const int64_t l = Ic[j * kc + c];
                    if (l < 0 || l == j) continue;                                      /* cache[i] without i, :168-169 */
                    total += (double)(1.0f - lo_dot(E_tr + l * (int64_t)d, qv + i * (int64_t)d, d));
                    ++count;
                }
            } else {
                for (int b = 0; b < kq; ++b) {
                    const int64_t l = Im[i * kq + b];
                    if (l < 0) continue;
                    total += (double)(1.0f - lo_dot(E_tr + j * (int64_t)d, E_tr + l * (int64_t)d, d));
                }
            }
        }
        out[i] = (method == 0) ? (float)(count ? total / (double)count : NAN) : (float)(total / ((double)k * (double)k));
    }
    free(Dc); free(Ic); free(Dm); free(Im);
}

int lo_has_simd(void) { return LO_SIMD; }

#ifdef _OPENMP
#include <omp.h>
void lo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int lo_get_threads(void) { return omp_get_max_threads(); }
#else
void lo_set_threads(int n) { (void)n; }
int lo_get_threads(void) { return 1; }
#endif
