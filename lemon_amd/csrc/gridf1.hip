// Hyper-parameter grid of the LEMoN score, evaluated as ONE batch on the GPU (gfx950 only).
//
// run_lemon.py:319-384 -> maximize_metric (lib/metrics/utils.py:151-196) evaluates, for each of the
// 21*21*4*4 = 7056 grid points, scores = calc_scores_given_hparams_vectorized(val, hp) (:47-82) and
// optimize_f1_efficient(y, scores) (:286-296) = scipy.optimize.fminbound on t -> -F1(y, scores >= t)
// with xtol = 1e-8.  On the host that is one K5 launch + one device->host copy + ~35 numpy F1
// evaluations per grid point.  Here: kernel 1 fills scores[G, N] with exactly K5's arithmetic
// (float64, same expression order), kernel 2 runs one bounded-Brent search per grid point -- one
// wavefront each, a faithful restatement of scipy's _minimize_scalar_bounded in IEEE double (no FMA
// contraction inside the search; the score kernel keeps the default so that it matches k_score), F1 from integer counts -- and returns (F1, threshold) per grid point,
// bit-identical to the host path (tests/test_gpu_parity.py).  A non-finite score row yields F1 = 0, as the
// host objective does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void k_grid_scores(const float *__restrict__ d1, const float *__restrict__ Dn,
                                                     const float *__restrict__ trn, const float *__restrict__ dn,
                                                     const float *__restrict__ Dm, const float *__restrict__ trm,
                                                     const float *__restrict__ dm, int64_t n, int k,
                                                     const double *__restrict__ hp /* [G,6] */, double *__restrict__ score) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = blockIdx.y;
    const double beta = hp[6 * g], gamma = hp[6 * g + 1], t1n = hp[6 * g + 2], t2n = hp[6 * g + 3], t1m = hp[6 * g + 4],
                 t2m = hp[6 * g + 5];
    double sn = 0.0, sm = 0.0;
    for (int j = 0; j < k; ++j) {                        // identical to k_score (rowwise.hip)
        const int64_t o = i * (int64_t)k + j;
        sn += exp(-t1n * (double)Dn[o]) * exp(-t2n * (double)trn[o]) * (double)dn[o];
        sm += exp(-t1m * (double)Dm[o]) * exp(-t2m * (double)trm[o]) * (double)dm[o];
    }
    const double a = sn / (double)k, b = sm / (double)k;
    score[(int64_t)g * n + i] = (double)d1[i] + beta * a + gamma * b;
}

struct F1Ctx { const double *s; const uint8_t *y; int64_t n; int64_t pos; int lane; };

// -F1(y, s >= t): sklearn's binary F1 from integer counts, 0 when nothing is counted
__device__ __forceinline__ double neg_f1(const F1Ctx &c, double t) {
#pragma clang fp contract(off)
    long long pred = 0, tp = 0;
    for (int64_t i = c.lane; i < c.n; i += 64) {
        const bool p = c.s[i] >= t;
        pred += p;
        tp += p && c.y[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { pred += __shfl_xor(pred, off); tp += __shfl_xor(tp, off); }
    const long long denom = 2 * tp + (pred - tp) + (c.pos - tp);
    return denom ? -(2.0 * (double)tp / (double)denom) : -0.0;
}

__device__ __forceinline__ double sgn1(double v) { return v < 0.0 ? -1.0 : 1.0; }   // np.sign(v) + (v == 0)

__global__ __launch_bounds__(256) void k_grid_brent(const double *__restrict__ score, const uint8_t *__restrict__ y,
                                                    int64_t n, int G, double xatol, int maxfun, double sqrt_eps,
                                                    double golden_mean, double *__restrict__ out_f1,
                                                    double *__restrict__ out_thres) {
#pragma clang fp contract(off)       // the Brent iteration must round like the interpreter: one operation at a time
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;                                  // wave-uniform
    F1Ctx c;
    c.s = score + (int64_t)g * n; c.y = y; c.n = n; c.lane = lane;
    // bounds = (min, max) of the row; positives; finiteness
    double lo = INFINITY, hi = -INFINITY;
    long long pos = 0; int bad = 0;
    for (int64_t i = lane; i < n; i += 64) {
        const double v = c.s[i];
        bad |= !(fabs(v) <= 1.7976931348623157e308);
        lo = v < lo ? v : lo; hi = v > hi ? v : hi;
        pos += y[i] != 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
        pos += __shfl_xor(pos, off); bad |= __shfl_xor(bad, off);
    }
    c.pos = pos;
    if (bad) { if (lane == 0) { out_f1[g] = 0.0; out_thres[g] = __builtin_nan(""); } return; }

    // scipy.optimize._optimize._minimize_scalar_bounded (scipy 1.15), every wave-uniform value in double
    double a = lo, b = hi;
    double fulc = a + golden_mean * (b - a);
    double nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf;
    double fx = neg_f1(c, x);
    int num = 1;
    double fu = INFINITY;
    double ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
    double tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        int golden = 1;
        if (fabs(e) > tol1) {                            // parabolic fit
            golden = 0;
            double r = (xf - nfc) * (fx - ffulc);
            double q = (xf - fulc) * (fx - fnfc);
            double p = (xf - fulc) * q - (xf - nfc) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            r = e;
            e = rat;
            if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
                rat = (p + 0.0) / q;
                x = xf + rat;
                if (((x - a) < tol2) || ((b - x) < tol2)) rat = tol1 * sgn1(xm - xf);
            } else {
                golden = 1;
            }
        }
        if (golden) {
            e = (xf >= xm) ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        const double ar = fabs(rat);
        x = xf + sgn1(rat) * (ar > tol1 ? ar : tol1);    // np.maximum(abs(rat), tol1); neither is NaN here
        fu = neg_f1(c, x);
        num += 1;
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc;
            nfc = xf; fnfc = fx;
            xf = x; fx = fu;
        } else {
            if (x < xf) a = x; else b = x;
            if ((fu <= fnfc) || (nfc == xf)) {
                fulc = nfc; ffulc = fnfc;
                nfc = x; fnfc = fu;
            } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
                fulc = x; ffulc = fu;
            }
        }
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
        tol2 = 2.0 * tol1;
        if (num >= maxfun) break;
    }
    const double best = -neg_f1(c, xf);                  // optimize_f1_efficient re-evaluates at the returned x
    if (lane == 0) { out_f1[g] = best; out_thres[g] = xf; }
}

}  // namespace

extern "C" int lemon_grid_f1(const float *d1_dev, const float *D_n_dev, const float *dists_tr_n_dev, const float *dists_n_dev,
                             const float *D_m_dev, const float *dists_tr_m_dev, const float *dists_m_dev,
                             const uint8_t *y_dev, int64_t n, int k, const double *hp_dev, int G, double xtol, int maxfun,
                             double *scores_ws_dev, double *f1_dev, double *thres_dev, void *stream_) {
    LEMON_REQUIRE(n > 0 && k > 0 && G >= 0 && maxfun >= 1, "n > 0, k > 0, G >= 0, maxfun >= 1");
    if (G == 0) return LEMON_OK;
    LEMON_REQUIRE(G <= 65535, "at most 65535 grid points per call");
    LEMON_REQUIRE(d1_dev && D_n_dev && dists_tr_n_dev && dists_n_dev && D_m_dev && dists_tr_m_dev && dists_m_dev && y_dev &&
                      hp_dev && scores_ws_dev && f1_dev && thres_dev, "null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_grid_scores, dim3((unsigned)((n + 255) / 256), (unsigned)G), dim3(256), 0, stream, d1_dev, D_n_dev,
                       dists_tr_n_dev, dists_n_dev, D_m_dev, dists_tr_m_dev, dists_m_dev, n, k, hp_dev, scores_ws_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    const double sqrt_eps = sqrt(2.2e-16), golden_mean = 0.5 * (3.0 - sqrt(5.0));
    hipLaunchKernelGGL(k_grid_brent, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, stream, scores_ws_dev, y_dev, n, G, xtol,
                       maxfun, sqrt_eps, golden_mean, f1_dev, thres_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
