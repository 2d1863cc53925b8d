// Linear layers of the CLIP towers with the element-wise work fused into the GEMM (gfx950 only).
//
//   y = act(alpha x W^T + bias) [+ residual]      x [m,k], W [n,k] (nn.Linear layout), y/residual [m,n], fp32
//
// The encoders' GEMMs (encode_image / encode_text, lib/models/downstream_models.py:30-41; MLP and
// attention projections of HF CLIPEncoderLayer / lib/models/chexzero_clip.py:191-212) are plain library
// GEMMs: they run on hipBLASLt.  What this file adds is (1) the epilogues PyTorch cannot express --
// SiLU (hipBLASLt's Swish epilogue; QuickGELU z*sigmoid(1.702z), chexzero_clip.py:186-188, is
// silu(1.702 z)/1.702: alpha = 1.702 here with a pre-scaled bias, alpha = 1/1.702 in the GEMM that
// consumes it) and the residual add as the GEMM's beta*C term -- which removes one read+write pass
// over the [m,3072] MLP activations and both residual-add passes per block, and (2) explicit solution selection: the
// library's default heuristic is ~15 % off the best solution for the ViT-B/32 shapes.
//
// Solution policy (reproducibility first: a different solution = a different fp32 summation order = different
// embedding bits):
//   * a key (m,n,k,epilogue,residual) found in the results file (lemon_amd/data/linear_gfx950.csv, written by the
//     OFFLINE tuner tools/tune_gemms.py) uses the recorded solution index.  The file is stamped with the hipBLASLt
//     version and the gfx arch it was made with and is ignored on a mismatch; each recorded index is checked ONCE per
//     process, on its first use, against the library's first-ranked solution on the caller's own operands
//     (|delta| <= 1e-3 max|y|) and dropped if it disagrees;
//   * any other key uses the library's first-ranked supported solution (hipblasLtMatmulAlgoGetHeuristic): no timing,
//     no allocation, no synchronisation in the inference path, and the same choice in every process;
//   * the timing race over all solutions runs only when tuning was switched on explicitly
//     (lemon_linear_set_tuning(1) or LEMON_LINEAR_TUNE=1).
// The hipBLASLt handle, its workspace and the solution cache are the library's only process-wide state (one set per
// device, guarded by a mutex).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "common.hpp"

namespace {

#define LT_CHECK(expr)                                                                              \
    do {                                                                                            \
        hipblasStatus_t st_ = (expr);                                                               \
        if (st_ != HIPBLAS_STATUS_SUCCESS) {                                                        \
            lemon_set_error("%s failed with hipblasStatus %d (%s:%d)", #expr, (int)st_, __FILE__, __LINE__); \
            return LEMON_E_HIP;                                                                     \
        }                                                                                           \
    } while (0)

typedef std::tuple<int64_t, int, int, int, int, int> LinKey;   // m, n, k, epilogue, has_residual, operand type (0 f32, 1 bf16, 2 fp16)

struct LinState {
    hipblasLtHandle_t handle = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    std::map<LinKey, hipblasLtMatmulAlgo_t> algo;     // the solution in use for the key in this process
    std::map<LinKey, int> index;                      // solution index (results file or tuner)
    std::map<LinKey, float> usec;
    std::map<LinKey, bool> from_file;                 // recorded index not yet checked against the first-ranked solution
    double tune_budget_ms = 6000.0;
    int version = 0;                                  // hipblasLtGetVersion
    char arch[64] = {0};                              // gcnArchName up to the first ':'
};
constexpr int LEMON_MAX_DEVICES = 16;
std::mutex g_mu;
LinState g_states[LEMON_MAX_DEVICES];
int g_tuning = -1;                                    // -1: read LEMON_LINEAR_TUNE on first use
#define g_lin (*g_cur)
thread_local LinState *g_cur = nullptr;               // state of the calling thread's current device (set under g_mu)

// validation of a candidate's output against the library's first-ranked solution: a solution that is
// merely fast but wrong for an odd shape must never be recorded
__global__ void k_absmax(const float *__restrict__ a, int64_t n, unsigned *__restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = fabsf(a[i]);
        m = (v > m || v != v) ? (v != v ? INFINITY : v) : m;
    }
    atomicMax(out, __float_as_uint(m));                  // non-negative floats order like their bit patterns
}
__global__ void k_count_off(const float *__restrict__ a, const float *__restrict__ b, int64_t n, float tol,
                            unsigned *__restrict__ out) {
    unsigned bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad += !(fabsf(a[i] - b[i]) <= tol);             // NaN counts as off
    if (bad) atomicAdd(out, bad);
}

// position independence of a candidate: with every row of x identical, every row of y must come out bit-identical.
// Stream-K style solutions cut the k loop differently from tile to tile, so the same prompt embedded at two positions of a
// micro-batch would differ in the last bit (bench.py: thousands of "distinct" embeddings for 100 class prompts); the tuner
// only records solutions that pass (LEMON_LINEAR_ALLOW_POSITION_DEPENDENT=1 lifts the requirement).
__global__ void k_replicate_row(const char *__restrict__ row0, int64_t rows, int64_t row_bytes, char *__restrict__ dst) {
    const int64_t total = rows * (row_bytes / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<unsigned *>(dst)[i] = reinterpret_cast<const unsigned *>(row0)[i % (row_bytes / 4)];
}
__global__ void k_rows_differ(const float *__restrict__ y, int64_t m, int n, unsigned *__restrict__ out) {
    unsigned bad = 0;
    const int64_t total = m * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        bad += __float_as_uint(y[i]) != __float_as_uint(y[i % n]);
    if (bad) atomicAdd(out, bad);
}

struct Problem {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    ~Problem() {
        if (la) (void)hipblasLtMatrixLayoutDestroy(la);
        if (lb) (void)hipblasLtMatrixLayoutDestroy(lb);
        if (lc) (void)hipblasLtMatrixLayoutDestroy(lc);
        if (desc) (void)hipblasLtMatmulDescDestroy(desc);
    }
};

// Row-major y[m,n] = x[m,k] W[n,k]^T is, in hipBLASLt's column-major terms, D[n,m] = op_T(A[k,n]) B[k,m]
// with A = W (ld k), B = x (ld k), C/D = residual/y (ld n); the bias runs along D's rows (n).
// dt = 1: A (weights) and B (activations) are bf16 -- the 6k-long concatenated split operands of lemon_linear_bf16x6 --,
// dt = 2: fp16, the 3k-long operands of lemon_linear_f16x3; C / D / bias / scale type stay fp32, the products accumulate in fp32
inline hipDataType operand_type(int dt) { return dt == 1 ? HIP_R_16BF : dt == 2 ? HIP_R_16F : HIP_R_32F; }
int make_problem(Problem &p, int64_t m, int n, int k, int epilogue, const float *bias, int dt = 0) {
    LT_CHECK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    if (dt) {
        const hipDataType bt = HIP_R_32F;
        LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    const hipblasLtEpilogue_t epi = (hipblasLtEpilogue_t)epilogue;
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
    if (bias) LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    const hipDataType ot = operand_type(dt);
    LT_CHECK(hipblasLtMatrixLayoutCreate(&p.la, ot, (uint64_t)k, (uint64_t)n, k));
    LT_CHECK(hipblasLtMatrixLayoutCreate(&p.lb, ot, (uint64_t)k, (uint64_t)m, k));
    LT_CHECK(hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, (uint64_t)n, (uint64_t)m, n));
    return LEMON_OK;
}

int ensure_state(hipStream_t stream) {
    int dev = 0;
    LEMON_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= LEMON_MAX_DEVICES) { lemon_set_error("device ordinal %d out of range", dev); return LEMON_E_INVALID; }
    g_cur = &g_states[dev];
    if (g_tuning < 0) {
        const char *e = getenv("LEMON_LINEAR_TUNE");
        g_tuning = (e && *e && strcmp(e, "0") != 0) ? 1 : 0;
    }
    if (!g_lin.handle) {
        LT_CHECK(hipblasLtCreate(&g_lin.handle));
        if (const char *e = getenv("LEMON_LINEAR_TUNE_MS")) g_lin.tune_budget_ms = atof(e);
        (void)hipblasLtGetVersion(g_lin.handle, &g_lin.version);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            size_t i = 0;
            for (; i + 1 < sizeof g_lin.arch && prop.gcnArchName[i] && prop.gcnArchName[i] != ':'; ++i) g_lin.arch[i] = prop.gcnArchName[i];
            g_lin.arch[i] = 0;
        }
    }
    if (!g_lin.ws) {
        g_lin.ws_bytes = (size_t)64 << 20;
        LEMON_HIP_CHECK(hipMalloc(&g_lin.ws, g_lin.ws_bytes));
    }
    (void)stream;
    return LEMON_OK;
}

// time one candidate on the caller's stream (D = scratch so that an in-place residual is not accumulated)
float time_algo(Problem &p, const hipblasLtMatmulAlgo_t &algo, const void *x, const void *w, const float *c,
                float beta, float *d, int reps, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    const float alpha = 1.0f;                            // timing only: the scale does not change the kernel
    if (hipblasLtMatmul(g_lin.handle, p.desc, &alpha, w, p.la, x, p.lb, &beta, c, p.lc, d, p.lc, &algo, g_lin.ws,
                        g_lin.ws_bytes, stream) != HIPBLAS_STATUS_SUCCESS)
        return -1.0f;
    if (reps <= 0) return 0.0f;                          // one launch, caller synchronises
    (void)hipEventRecord(e0, stream);
    for (int r = 0; r < reps; ++r)
        (void)hipblasLtMatmul(g_lin.handle, p.desc, &alpha, w, p.la, x, p.lb, &beta, c, p.lc, d, p.lc, &algo, g_lin.ws,
                              g_lin.ws_bytes, stream);
    (void)hipEventRecord(e1, stream);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.0f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

bool supported(Problem &p, hipblasLtMatmulAlgo_t &algo, float beta) {
    const float alpha = 1.0f;
    size_t need = 0;
    return hipblaslt_ext::matmulIsAlgoSupported(g_lin.handle, p.desc, &alpha, p.la, p.lb, &beta, p.lc, p.lc, algo, need) ==
               HIPBLAS_STATUS_SUCCESS &&
           need <= g_lin.ws_bytes;
}

// benchmark the solutions that support this problem; returns the winner (synchronises the stream)
int tune(const LinKey &key, Problem &p, const void *x, const void *w, const float *residual, int64_t m, int n,
         hipStream_t stream, hipblasLtMatmulAlgo_t *best_out) {
    const hipDataType ot = operand_type(std::get<5>(key));
    const float beta = residual ? 1.0f : 0.0f;
    float *scratch = nullptr;
    LEMON_HIP_CHECK(hipMalloc((void **)&scratch, (size_t)m * n * sizeof(float)));
    const float *c = residual ? residual : scratch;
    hipEvent_t e0, e1, t0, t1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    (void)hipEventRecord(t0, stream);

    std::vector<hipblasLtMatmulHeuristicResult_t> cand;
    {   // the library's own ranking first: a sane answer even if the budget runs out early
        hipblasLtMatmulPreference_t pref = nullptr;
        if (hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS) {
            (void)hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &g_lin.ws_bytes,
                                                        sizeof(g_lin.ws_bytes));
            hipblasLtMatmulHeuristicResult_t top[8];
            int got = 0;
            if (hipblasLtMatmulAlgoGetHeuristic(g_lin.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 8, top, &got) ==
                HIPBLAS_STATUS_SUCCESS)
                for (int i = 0; i < got; ++i) cand.push_back(top[i]);
            (void)hipblasLtMatmulPreferenceDestroy(pref);
        }
    }
    const size_t n_heur = cand.size();
    {
        std::vector<hipblasLtMatmulHeuristicResult_t> all;
        if (hipblaslt_ext::getAllAlgos(g_lin.handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N,
                                       ot, ot, HIP_R_32F, HIP_R_32F, HIPBLAS_COMPUTE_32F,
                                       all) == HIPBLAS_STATUS_SUCCESS)
            cand.insert(cand.end(), all.begin(), all.end());
    }
    std::vector<std::pair<float, size_t>> timed;        // (usec, candidate)
    for (size_t i = 0; i < cand.size(); ++i) {
        if (!supported(p, cand[i].algo, beta)) continue;
        const float us = time_algo(p, cand[i].algo, x, w, c, beta, scratch, 1, stream, e0, e1);
        if (us > 0.f) timed.push_back({us, i});
        if (i >= n_heur) {                               // budget applies to the exhaustive part only
            (void)hipEventRecord(t1, stream); (void)hipEventSynchronize(t1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, t0, t1);
            if (ms > g_lin.tune_budget_ms) break;
        }
    }
    int rc = LEMON_OK;
    float *ref = nullptr;
    unsigned *flag = nullptr;
    char *xrep = nullptr;
    size_t best = 0; float best_us = 1e30f; bool found = false; int rejected = 0, pos_dep = 0;
    if (!timed.empty() && (hipMalloc((void **)&ref, (size_t)m * n * sizeof(float)) != hipSuccess ||
                           hipMalloc((void **)&flag, 2 * sizeof(unsigned)) != hipSuccess)) {
        lemon_set_error("lemon_linear_f32: validation buffers (%lld x %d floats)", (long long)m, n);
        rc = LEMON_E_NOMEM;
    } else if (!timed.empty()) {
        // reference output: the first timed candidate in the library's own ranking order
        size_t ref_cand = timed[0].second;
        for (const auto &tc : timed) if (tc.second < ref_cand) ref_cand = tc.second;
        // (C = D = a zero-filled buffer when there is no residual: immune to a solution that reads C
        // although beta == 0; the candidates below are then run with a NaN-poisoned C to expose exactly that)
        (void)hipMemsetAsync(ref, 0, (size_t)m * n * sizeof(float), stream);
        (void)time_algo(p, cand[ref_cand].algo, x, w, residual ? residual : ref, beta, ref, 0, stream, e0, e1);
        const int64_t cnt = (int64_t)m * n;
        const unsigned blocks = (unsigned)std::min<int64_t>((cnt + 255) / 256, 2048);
        (void)hipMemsetAsync(flag, 0, 2 * sizeof(unsigned), stream);
        hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, stream, ref, cnt, flag);
        unsigned hflag[2] = {0, 0};
        (void)hipMemcpyAsync(hflag, flag, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        float ref_max; memcpy(&ref_max, &hflag[0], sizeof(float));
        const float tol = 1e-3f * ref_max + 1e-6f;
        std::sort(timed.begin(), timed.end());
        int finals = 0;
        const bool need_pos_indep = !(getenv("LEMON_LINEAR_ALLOW_POSITION_DEPENDENT") && atoi(getenv("LEMON_LINEAR_ALLOW_POSITION_DEPENDENT")));
        const int64_t row_bytes = (int64_t)std::get<2>(key) * (std::get<5>(key) ? 2 : 4);
        if (need_pos_indep && m > 1 && (row_bytes & 3) == 0 && hipMalloc((void **)&xrep, (size_t)m * row_bytes) == hipSuccess)
            hipLaunchKernelGGL(k_replicate_row, dim3(2048), dim3(256), 0, stream, reinterpret_cast<const char *>(x), m, row_bytes, xrep);
        for (size_t j = 0; j < timed.size() && finals < 6; ++j) {   // front runners: validate, then re-time properly
            const size_t ci = timed[j].second;
            if (xrep) {      // identical rows in, identical rows out (no residual: C = D = scratch, beta as the problem has it is irrelevant here)
                const float zero = 0.0f, one = 1.0f;
                (void)hipMemsetAsync(scratch, 0, (size_t)m * n * sizeof(float), stream);
                if (hipblasLtMatmul(g_lin.handle, p.desc, &one, w, p.la, xrep, p.lb, &zero, scratch, p.lc, scratch, p.lc, &cand[ci].algo,
                                    g_lin.ws, g_lin.ws_bytes, stream) != HIPBLAS_STATUS_SUCCESS) { ++rejected; continue; }
                (void)hipMemsetAsync(flag + 1, 0, sizeof(unsigned), stream);
                hipLaunchKernelGGL(k_rows_differ, dim3(blocks), dim3(256), 0, stream, scratch, m, n, flag + 1);
                (void)hipMemcpyAsync(&hflag[1], flag + 1, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
                (void)hipStreamSynchronize(stream);
                if (hflag[1] != 0) { ++rejected; ++pos_dep; continue; }
            }
            {
                if (!residual) (void)hipMemsetAsync(scratch, 0xFF, (size_t)m * n * sizeof(float), stream);   // NaN
                (void)time_algo(p, cand[ci].algo, x, w, c, beta, scratch, 0, stream, e0, e1);
                (void)hipMemsetAsync(flag + 1, 0, sizeof(unsigned), stream);
                hipLaunchKernelGGL(k_count_off, dim3(blocks), dim3(256), 0, stream, scratch, ref, cnt, tol, flag + 1);
                (void)hipMemcpyAsync(&hflag[1], flag + 1, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
                (void)hipStreamSynchronize(stream);
                if (hflag[1] != 0 || !(ref_max < INFINITY)) { ++rejected; continue; }
            }
            ++finals;
            const float us = time_algo(p, cand[ci].algo, x, w, c, beta, scratch, 5, stream, e0, e1);
            if (us > 0.f && us < best_us) { best_us = us; best = ci; found = true; }
        }
    }
    if (xrep) (void)hipFree(xrep);
    if (rc == LEMON_OK && !found) {
        lemon_set_error("lemon_linear_f32: no hipBLASLt solution supports m=%lld n=%d k=%d epilogue=%d", (long long)m, n,
                        std::get<2>(key), std::get<3>(key));
        rc = LEMON_E_INVALID;
    } else if (rc == LEMON_OK) {
        *best_out = cand[best].algo;
        g_lin.index[key] = hipblaslt_ext::getIndexFromAlgo(cand[best].algo);
        g_lin.usec[key] = best_us;
        if (getenv("LEMON_LINEAR_VERBOSE"))
            fprintf(stderr, "[lemon_linear] m=%lld n=%d k=%d epi=%d res=%d dt=%d: %zu of %zu solutions timed, %d rejected by validation (%d position-dependent), best index %d %.1f us\n",
                    (long long)m, n, std::get<2>(key), std::get<3>(key), std::get<4>(key), std::get<5>(key), timed.size(), cand.size(), rejected,
                    pos_dep, g_lin.index[key], best_us);
    }
    if (ref) (void)hipFree(ref);
    if (flag) (void)hipFree(flag);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    (void)hipFree(scratch);
    return rc;
}

// the library's own first choice among the solutions that support the problem (deterministic, no timing)
int first_ranked(Problem &p, float beta, hipblasLtMatmulAlgo_t *out) {
    hipblasLtMatmulPreference_t pref = nullptr;
    LT_CHECK(hipblasLtMatmulPreferenceCreate(&pref));
    (void)hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &g_lin.ws_bytes,
                                                sizeof(g_lin.ws_bytes));
    hipblasLtMatmulHeuristicResult_t top[8];
    int got = 0;
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_lin.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 8, top, &got);
    (void)hipblasLtMatmulPreferenceDestroy(pref);
    if (st == HIPBLAS_STATUS_SUCCESS)
        for (int i = 0; i < got; ++i)
            if (supported(p, top[i].algo, beta)) { *out = top[i].algo; return LEMON_OK; }
    lemon_set_error("lemon_linear_f32: hipBLASLt offers no solution for this problem (heuristic status %d, %d candidates)", (int)st, got);
    return LEMON_E_INVALID;
}

// one-time check of a recorded solution index on the caller's operands: its output must agree with the first-ranked
// solution's (synchronises the stream; runs once per key and process).  Returns 1 = agrees, 0 = rejected, <0 error.
int recorded_agrees(Problem &p, const hipblasLtMatmulAlgo_t &rec, const hipblasLtMatmulAlgo_t &first, const void *x,
                    const void *w, const float *residual, float alpha, int64_t m, int n, hipStream_t stream) {
    const float beta = residual ? 1.0f : 0.0f;
    const int64_t cnt = (int64_t)m * n;
    float *buf = nullptr;
    unsigned *flag = nullptr;
    if (hipMalloc((void **)&buf, (size_t)2 * cnt * sizeof(float)) != hipSuccess || hipMalloc((void **)&flag, 2 * sizeof(unsigned)) != hipSuccess) {
        if (buf) (void)hipFree(buf);
        lemon_set_error("lemon_linear_f32: validation buffers (2 x %lld x %d floats)", (long long)m, n);
        return LEMON_E_NOMEM;
    }
    float *ya = buf, *yb = buf + cnt;
    (void)hipMemsetAsync(ya, 0, (size_t)cnt * sizeof(float), stream);
    (void)hipMemsetAsync(yb, 0xFF, (size_t)cnt * sizeof(float), stream);      // NaN-poisoned C for the candidate (beta == 0 must not read it)
    (void)hipMemsetAsync(flag, 0, 2 * sizeof(unsigned), stream);
    bool ok = hipblasLtMatmul(g_lin.handle, p.desc, &alpha, w, p.la, x, p.lb, &beta, residual ? residual : ya, p.lc, ya, p.lc, &first,
                              g_lin.ws, g_lin.ws_bytes, stream) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && hipblasLtMatmul(g_lin.handle, p.desc, &alpha, w, p.la, x, p.lb, &beta, residual ? residual : yb, p.lc, yb, p.lc, &rec,
                               g_lin.ws, g_lin.ws_bytes, stream) == HIPBLAS_STATUS_SUCCESS;
    unsigned h[2] = {0, 1};
    if (ok) {
        const unsigned blocks = (unsigned)std::min<int64_t>((cnt + 255) / 256, 2048);
        hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, stream, ya, cnt, flag);
        (void)hipMemcpyAsync(h, flag, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        float ref_max; memcpy(&ref_max, &h[0], sizeof(float));
        hipLaunchKernelGGL(k_count_off, dim3(blocks), dim3(256), 0, stream, yb, ya, cnt, 1e-3f * ref_max + 1e-6f, flag + 1);
        (void)hipMemcpyAsync(&h[1], flag + 1, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        if (!(ref_max < INFINITY)) h[1] = 0;              // non-finite inputs: nothing to compare, keep the record
    }
    (void)hipFree(buf); (void)hipFree(flag);
    return ok && h[1] == 0 ? 1 : 0;
}

// y <- y/2 (1 + erf(y / sqrt 2)), four values per thread (the tail one by one)
__global__ __launch_bounds__(256) void k_gelu_inplace(float *__restrict__ y, int64_t total) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < total && (((uintptr_t)y) & 15) == 0) {
        float4 v = *reinterpret_cast<float4 *>(y + i);
        v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f)); v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
        v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f)); v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
        *reinterpret_cast<float4 *>(y + i) = v;
    } else {
        for (int64_t j = i; j < total && j < i + 4; ++j) y[j] = 0.5f * y[j] * (1.0f + erff(y[j] * 0.70710678118654752f));
    }
}

}  // namespace

extern "C" int lemon_linear_set_tuning(int enabled) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_tuning = enabled ? 1 : 0;
    return LEMON_OK;
}

static int linear_impl(int dt, const void *x_dev, const void *w_dev, const float *bias_dev, const float *residual_dev,
                       int64_t m, int n, int k, float alpha, int act, float *y_dev, void *stream_) {
    LEMON_REQUIRE(m >= 0 && n > 0 && k > 0, "m >= 0, n > 0, k > 0");
    LEMON_REQUIRE(act == LEMON_ACT_NONE || act == LEMON_ACT_SILU || act == LEMON_ACT_GELU, "act must be LEMON_ACT_NONE, _SILU or _GELU");
    LEMON_REQUIRE(!(act != LEMON_ACT_NONE && residual_dev), "activation and residual cannot be combined");
    if (m == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev && w_dev && y_dev, "null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = ensure_state(stream);
    if (rc) return rc;
    int epilogue = HIPBLASLT_EPILOGUE_DEFAULT;
    if (act == LEMON_ACT_SILU) epilogue = bias_dev ? HIPBLASLT_EPILOGUE_SWISH_BIAS_EXT : HIPBLASLT_EPILOGUE_SWISH_EXT;
    else if (bias_dev) epilogue = HIPBLASLT_EPILOGUE_BIAS;
    const LinKey key(m, n, k, epilogue, residual_dev ? 1 : 0, dt);
    const float beta = residual_dev ? 1.0f : 0.0f;
    Problem p;
    rc = make_problem(p, m, n, k, epilogue, bias_dev, dt);
    if (rc) return rc;

    hipblasLtMatmulAlgo_t algo;
    bool have = false;
    auto it = g_lin.algo.find(key);
    if (it != g_lin.algo.end()) {
        algo = it->second;
        have = supported(p, algo, beta);                 // also binds the problem to the algo struct
    }
    if (!have) {
        auto ix = g_lin.index.find(key);
        if (ix != g_lin.index.end() && g_lin.from_file.count(key)) {     // recorded index: check it once, then trust it
            std::vector<int> idx{ix->second};
            std::vector<hipblasLtMatmulHeuristicResult_t> res;
            hipblasLtMatmulAlgo_t first;
            if (hipblaslt_ext::getAlgosFromIndex(g_lin.handle, idx, res) == HIPBLAS_STATUS_SUCCESS && !res.empty() &&
                supported(p, res[0].algo, beta) && first_ranked(p, beta, &first) == LEMON_OK) {
                const int ok = recorded_agrees(p, res[0].algo, first, x_dev, w_dev, residual_dev, alpha, m, n, stream);
                if (ok < 0) return ok;
                if (ok == 1) { algo = res[0].algo; have = supported(p, algo, beta); }
                else if (getenv("LEMON_LINEAR_VERBOSE"))
                    fprintf(stderr, "[lemon_linear] recorded solution %d for m=%lld n=%d k=%d epi=%d disagrees with the first-ranked one: dropped\n",
                            ix->second, (long long)m, n, k, epilogue);
            }
            g_lin.from_file.erase(key);
            if (!have) { g_lin.index.erase(key); g_lin.usec.erase(key); }
        }
    }
    if (!have && g_tuning == 1) {                        // explicit offline tuning only
        rc = tune(key, p, x_dev, w_dev, residual_dev, m, n, stream, &algo);
        if (rc) return rc;
        have = supported(p, algo, beta);
        if (!have) { lemon_set_error("lemon_linear_f32: tuned solution rejected"); return LEMON_E_HIP; }
    }
    if (!have) {
        rc = first_ranked(p, beta, &algo);
        if (rc) return rc;
    }
    g_lin.algo[key] = algo;
    const float *c = residual_dev ? residual_dev : y_dev;
    LT_CHECK(hipblasLtMatmul(g_lin.handle, p.desc, &alpha, w_dev, p.la, x_dev, p.lb, &beta, c, p.lc, y_dev, p.lc, &algo,
                             g_lin.ws, g_lin.ws_bytes, stream));
    if (act == LEMON_ACT_GELU) {
        // exact (erf) GELU as one in-place pass behind the bias epilogue: the library's GELU epilogue is the tanh approximation
        const int64_t total = m * (int64_t)n;
        hipLaunchKernelGGL(k_gelu_inplace, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, stream, y_dev, total);
        LEMON_HIP_CHECK(hipGetLastError());
    }
    return LEMON_OK;
}

extern "C" int lemon_linear_f32(const float *x_dev, const float *w_dev, const float *bias_dev, const float *residual_dev,
                                int64_t m, int n, int k, float alpha, int act, float *y_dev, void *stream_) {
    return linear_impl(0, x_dev, w_dev, bias_dev, residual_dev, m, n, k, alpha, act, y_dev, stream_);
}

// fp32-equivalent linear layer on the bf16 matrix cores: x6 [m, 6k] and w6 [n, 6k] are the 3-way bf16 splits of the fp32
// operands (x = hi + mid + lo, each part bf16; lemon_split3_f32 / lemon_layernorm_split3) laid out so that ONE bf16 GEMM over
// the 6k-long k axis sums the six cross products of order <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi) in fp32:
//     x6 row = [hi | hi | mid | hi | mid | lo],   w6 row = [hi | mid | hi | lo | mid | hi].
// The dropped products are O(2^-24) of the result; the delivered error is bounded by the fp32 accumulation in the matrix pipe,
// as for the fp32 GEMM (max error vs float64 relative to the largest output 1.3-2.1e-6 at the tower shapes, fp32 GEMM 1.0-2.1e-6;
// tools/split_gemm_probe.py), at 1.4x the fp32 GEMM's speed.
// k6 = 6 k.  Same epilogues, same solution policy (keys carry the operand type) as lemon_linear_f32.
extern "C" int lemon_linear_bf16x6(const uint16_t *x6_dev, const uint16_t *w6_dev, const float *bias_dev, const float *residual_dev,
                                   int64_t m, int n, int k6, float alpha, int act, float *y_dev, void *stream_) {
    LEMON_REQUIRE(k6 % 6 == 0, "k6 must be 6 * k");
    return linear_impl(1, x6_dev, w6_dev, bias_dev, residual_dev, m, n, k6, alpha, act, y_dev, stream_);
}

// fp32-equivalent linear layer on the fp16 matrix cores with HALF the products of lemon_linear_bf16x6: x3 [m, 3k] and w3 [n, 3k]
// are the 2-way fp16 splits (lemon_split_f16x3 / lemon_layernorm_f16x3 / lemon_attention_f16x3):
//     x3 row = [hi | hi | lo 2^11],   w3 row = [hi | lo | hi 2^-11]  of  w * wscale,
// so ONE fp16 GEMM over the 3k-long k axis sums hi.hi + hi.lo + lo.hi in fp32; the caller passes alpha / wscale as `alpha`.
// 22 significant bits + the sign of lo per operand and a dropped lo.lo term of <= 2^-22 (typically 2^-26) of a product: against
// float64 the result is as accurate as the fp32 GEMM's (tools/split_gemm_probe.py).  Operand values beyond the fp16 range
// (|x| >= 65 520) turn the affected outputs into NaN.  k3 = 3 k.
extern "C" int lemon_linear_f16x3(const uint16_t *x3_dev, const uint16_t *w3_dev, const float *bias_dev, const float *residual_dev,
                                  int64_t m, int n, int k3, float alpha, int act, float *y_dev, void *stream_) {
    LEMON_REQUIRE(k3 % 3 == 0, "k3 must be 3 * k");
    return linear_impl(2, x3_dev, w3_dev, bias_dev, residual_dev, m, n, k3, alpha, act, y_dev, stream_);
}

// Results file: first line "# lemon_linear hipblaslt=<int> arch=<name>", then m,n,k,epilogue,residual,index,usec rows.
// Returns the number of keys taken (0 when the stamp does not match this process's library / device), <0 on error.
extern "C" int lemon_linear_load_tuned(const char *path) {
    LEMON_REQUIRE(path != nullptr, "path");
    FILE *f = fopen(path, "r");
    if (!f) { lemon_set_error("cannot open %s", path); return LEMON_E_INVALID; }
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = ensure_state(nullptr);
    if (rc) { fclose(f); return rc; }
    char line[512];
    int loaded = 0, version = -1;
    char arch[64] = {0};
    bool stamped = false;
    while (fgets(line, sizeof line, f)) {
        if (!stamped && sscanf(line, "# lemon_linear hipblaslt=%d arch=%63s", &version, arch) == 2) {
            stamped = true;
            if (version != g_lin.version || strcmp(arch, g_lin.arch) != 0) {
                if (getenv("LEMON_LINEAR_VERBOSE"))
                    fprintf(stderr, "[lemon_linear] %s was tuned for hipblaslt=%d arch=%s, this process has hipblaslt=%d arch=%s: ignored\n",
                            path, version, arch, g_lin.version, g_lin.arch);
                break;
            }
            continue;
        }
        long long m; int n, k, epi, res, index, dt = 0; float us;
        if (stamped && sscanf(line, "%lld,%d,%d,%d,%d,%d,%f,%d", &m, &n, &k, &epi, &res, &index, &us, &dt) >= 7) {
            const LinKey key((int64_t)m, n, k, epi, res, dt);
            if (!g_lin.algo.count(key)) { g_lin.index[key] = index; g_lin.usec[key] = us; g_lin.from_file[key] = true; ++loaded; }
        }
    }
    fclose(f);
    return loaded;
}

extern "C" int lemon_linear_dump_tuned(const char *path) {
    LEMON_REQUIRE(path != nullptr, "path");
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = ensure_state(nullptr);
    if (rc) return rc;
    FILE *f = fopen(path, "w");
    if (!f) { lemon_set_error("cannot write %s", path); return LEMON_E_INVALID; }
    fprintf(f, "# lemon_linear hipblaslt=%d arch=%s\n", g_lin.version, g_lin.arch);
    fprintf(f, "# m,n,k,epilogue,residual,hipblaslt_solution_index,usec[,operand type: 1 = bf16 split operands, k = 6 x the layer's k; 2 = fp16 split operands, k = 3 x]   (y = act(x W^T + b) [+ residual])\n");
    int rows = 0;
    for (const auto &kv : g_lin.index) {
        const LinKey &key = kv.first;
        fprintf(f, "%lld,%d,%d,%d,%d,%d,%.2f", (long long)std::get<0>(key), std::get<1>(key), std::get<2>(key),
                std::get<3>(key), std::get<4>(key), kv.second, g_lin.usec.count(key) ? g_lin.usec[key] : 0.0f);
        if (std::get<5>(key)) fprintf(f, ",%d", std::get<5>(key));
        fprintf(f, "\n");
        ++rows;
    }
    fclose(f);
    return rows;
}

// hipBLASLt version / arch stamp of the calling thread's current device (for the results file and for logs)
extern "C" int lemon_linear_stamp(int *hipblaslt_version, char *arch, int arch_len) {
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = ensure_state(nullptr);
    if (rc) return rc;
    if (hipblaslt_version) *hipblaslt_version = g_lin.version;
    if (arch && arch_len > 0) { strncpy(arch, g_lin.arch, (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
    return LEMON_OK;
}
