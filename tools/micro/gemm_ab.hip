// gemm_ab.hip -- same-process A/B of the two matrix-instruction shapes of the PRODUCT kernel (lemon_amd/csrc/gemm_f16x3.hip is
// included as source: the kernels timed here are the shipped ones, through the shipped C entry point).
//   k_gemm_f16x3t    v_mfma_f32_32x32x16_f16, double-buffered fragments, one barrier per k16 step
//   k_gemm_f16x3t16  v_mfma_f32_16x16x32_f16, single-buffered fragments, two barriers per k32 step
// Random operands (fp32 activations and weights split by the product's own packers), ROUNDS interleaved rounds per shape
// (cdna_hip_programming.md rule 24: perf deltas from interleaved rounds in one process), each round = REPS back-to-back
// launches between two HIP events; median and min per variant; results of both checked against a float64 reference on
// sampled outputs.  Shapes: the four block GEMMs of ViT-B/32 at the headline micro-batch (2 620 images x 50 tokens).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm tools/micro/gemm_ab.hip -o .variants/gemm_ab && .variants/gemm_ab
#include "../../lemon_amd/csrc/gemm_f16x3.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

static char g_err[512];
void lemon_set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LCHECK(x) do { int rc__ = (x); if (rc__ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc__, g_err); exit(1); } } while (0)

__global__ void k_fill_normal(float *g, size_t n, float scale, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned r = ((unsigned)i + seed) * 2654435761u; r ^= r >> 15; r *= 2246822519u; r ^= r >> 13; r *= 3266489917u; r ^= r >> 16;
        const float u = ((r & 255) + ((r >> 8) & 255) + ((r >> 16) & 255) + (r >> 24)) * (1.0f / 255.0f) - 2.0f;     // ~ N(0, 0.58)
        g[i] = u * scale;
    }
}
// fp32 [rows, k] -> tile-major activation operand (what lemon_layernorm_f16x3t / the fc1 epilogue write)
__global__ void k_pack_act(const float *__restrict__ x, int64_t rows, int k, unsigned short *__restrict__ at) {
    const int nch = k >> 3;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * nch) return;
    const int64_t r = t / nch;
    const int c = (int)(t - r * nch);
    const float4 *src = reinterpret_cast<const float4 *>(x + r * (int64_t)k + 8 * c);
    lemon_split::store_tiled8<lemon_split::TILE_A_ROWS, false>(at, r, k, c, src[0], src[1]);
}


// debug mode: x[m][k] = 64 m + k, W[n][k] = (k == n % K): out[m][n] = 64 m + n % K exactly; prints what arrived instead
__global__ void k_fill_dbg(float *x, float *w, int M, int N, int K) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M * K) x[t] = (float)(64 * (t / K) + t % K);
    if (t < N * K) w[t] = (t % K == (t / K) % K) ? 1.0f : 0.0f;
}
static int debug_run(int K) {
    const int M = 128, N = 256;
    float *x, *w, *y; uint16_t *at, *wt;
    CHECK(hipMalloc(&x, M * K * 4)); CHECK(hipMalloc(&w, N * K * 4)); CHECK(hipMalloc(&y, M * N * 4));
    CHECK(hipMalloc(&at, M * K * 4)); CHECK(hipMalloc(&wt, N * K * 4));
    hipLaunchKernelGGL(k_fill_dbg, dim3((N * K + 255) / 256), dim3(256), 0, 0, x, w, M, N, K);
    LCHECK(lemon_pack_weight_f16x3t(w, N, K, 1.0f, wt, nullptr));
    hipLaunchKernelGGL(k_pack_act, dim3((M * (K >> 3) + 255) / 256), dim3(256), 0, 0, x, (int64_t)M, K, at);
    for (int v = 0; v < 2; ++v) {
        g_mfma_shape = v ? 16 : 32;
        CHECK(hipMemset(y, 0, M * N * 4));
        LCHECK(lemon_linear_f16x3t(at, wt, nullptr, nullptr, M, N, K, 1.0f, LEMON_ACT_NONE, 0, y, nullptr));
        CHECK(hipDeviceSynchronize());
        std::vector<float> h(M * N);
        CHECK(hipMemcpy(h.data(), y, M * N * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        int tile_bad[8][16] = {};
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                const float want = 64.0f * m + n % K;
                if (h[m * N + n] != want) { ++bad; ++tile_bad[m / 16][n / 16]; }
            }
        if (bad) {
            printf("  wrong outputs per 16 x 16 tile (rows: m / 16, columns: n / 16):\n");
            for (int i = 0; i < 8; ++i) { printf("   "); for (int j = 0; j < 16; ++j) printf(" %3d", tile_bad[i][j]); printf("\n"); }
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 16; ++j)
                    if (tile_bad[i][j] && ((i == 4 && (j == 0 || j == 7)) || (i == 5 && j == 0) || (i==0 && j==0))) {
                        printf("  tile m %d.. n %d.. got - want (rows m, columns n):\n", 16 * i, 16 * j);
                        for (int a = 0; a < 16; ++a) { printf("   "); for (int b = 0; b < 16; ++b) printf(" %8g", h[(16 * i + a) * N + 16 * j + b] - (64.0f * (16 * i + a) + (16 * j + b) % K)); printf("\n"); }
                    }
        }
        printf("debug mfma %d K=%d: %d of %d wrong\n", v ? 16 : 32, K, bad, M * N);
    }
    return 0;
}

struct Shape { const char *name; int m, n, k, epi; };

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "debug")) { debug_run(32); debug_run(64); debug_run(128); return 0; }
    const bool walk = argc > 1 && !strcmp(argv[1], "walk");
    if (walk) { --argc; ++argv; }
    if (getenv("GEMM_GM")) { g_gm = atoi(getenv("GEMM_GM")); g_gn = atoi(getenv("GEMM_GN") ? getenv("GEMM_GN") : "1"); }
    const int rounds = argc > 1 ? atoi(argv[1]) : 7, reps = argc > 2 ? atoi(argv[2]) : 10;
    const int M = argc > 3 ? atoi(argv[3]) : 131000;
    const Shape shapes[] = {{"fc1 (SiLU -> operand)", M, 3072, 768, 1}, {"fc2 (+residual)", M, 768, 3072, 0},
                            {"qkv", M, 2304, 768, 0}, {"out-proj (+residual)", M, 768, 768, 0},
                            {"text fc1 m=40000", 40000, 2048, 512, 1}, {"text fc2 m=40000", 40000, 512, 2048, 0}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int bad = 0;
    for (const Shape &s : shapes) {
        const int64_t mp = (s.m + 127) / 128 * 128;
        float *x, *w, *bias, *res; uint16_t *at, *wt; void *out[2];
        CHECK(hipMalloc(&x, (size_t)s.m * s.k * 4)); CHECK(hipMalloc(&w, (size_t)s.n * s.k * 4));
        CHECK(hipMalloc(&bias, (size_t)s.n * 4)); CHECK(hipMalloc(&res, (size_t)s.m * s.n * 4));
        CHECK(hipMalloc(&at, (size_t)mp * s.k * 4)); CHECK(hipMalloc(&wt, (size_t)s.n * s.k * 4));
        const size_t out_bytes = s.epi ? (size_t)mp * s.n * 4 : (size_t)s.m * s.n * 4;
        CHECK(hipMalloc(&out[0], out_bytes)); CHECK(hipMalloc(&out[1], out_bytes));
        CHECK(hipMemset(at, 0, (size_t)mp * s.k * 4));
        hipLaunchKernelGGL(k_fill_normal, dim3(4096), dim3(256), 0, 0, x, (size_t)s.m * s.k, getenv("GEMM_ZERO") ? 0.0f : 1.0f, 1u);
        hipLaunchKernelGGL(k_fill_normal, dim3(4096), dim3(256), 0, 0, w, (size_t)s.n * s.k, getenv("GEMM_ZERO") ? 0.0f : 0.02f, 77u);
        hipLaunchKernelGGL(k_fill_normal, dim3(64), dim3(256), 0, 0, bias, (size_t)s.n, 0.1f, 5u);
        hipLaunchKernelGGL(k_fill_normal, dim3(4096), dim3(256), 0, 0, res, (size_t)s.m * s.n, 1.0f, 9u);
        const float wscale = 8192.0f * 16.0f;     // max |w| ~ 0.04 -> ~2^12..2^13 (a power of two, as weight_scale_f16x3 picks)
        LCHECK(lemon_pack_weight_f16x3t(w, s.n, s.k, wscale, wt, nullptr));
        {
            const int64_t threads = (int64_t)s.m * (s.k >> 3);
            hipLaunchKernelGGL(k_pack_act, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, x, (int64_t)s.m, s.k, at);
        }
        CHECK(hipDeviceSynchronize());
        const float alpha = 1.0f / wscale;
        // GEMM_LN=fold | emit: variant 1 is the LayerNorm-fold form of the 16x16x32 kernel (identity row affine / scratch emit
        // buffers: the same result), variant 0 the plain 16x16x32 kernel -- the cost of the fold's epilogues
        const char *ln = getenv("GEMM_LN");
        const bool ln_fold = ln && !strcmp(ln, "fold"), ln_emit = ln && !strcmp(ln, "emit") && !s.epi;
        static float *aff = nullptr, *csum = nullptr, *stats = nullptr; static uint16_t *emit_t = nullptr;
        if (ln && !aff) {
            CHECK(hipMalloc(&aff, (size_t)131072 * 8)); CHECK(hipMalloc(&csum, 4096 * 4)); CHECK(hipMalloc(&stats, (size_t)131072 * 32 * 8));
            CHECK(hipMalloc(&emit_t, (size_t)131072 * 3072 * 4));
            std::vector<float> h(131072 * 2);
            for (size_t i = 0; i < h.size(); i += 2) { h[i] = 1.0f; h[i + 1] = 0.0f; }
            CHECK(hipMemcpy(aff, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemset(csum, 0, 4096 * 4));
        }
        auto run = [&](int v) {
            g_mfma_shape = (v || ln) ? 16 : 32;
            const float *resid_p = (s.epi || getenv("GEMM_NO_RESIDUAL")) ? nullptr : res;
            if (v && ln_fold) LCHECK(lemon_linear_f16x3t_ln(at, wt, bias, resid_p, s.m, s.n, s.k, alpha, s.epi ? LEMON_ACT_SILU : LEMON_ACT_NONE, s.epi, out[v], aff, csum, nullptr, nullptr, nullptr));
            else if (v && ln_emit) LCHECK(lemon_linear_f16x3t_ln(at, wt, bias, resid_p, s.m, s.n, s.k, alpha, LEMON_ACT_NONE, 0, out[v], nullptr, nullptr, emit_t, stats, nullptr));
            else LCHECK(lemon_linear_f16x3t(at, wt, bias, resid_p, s.m, s.n, s.k, alpha, s.epi ? LEMON_ACT_SILU : LEMON_ACT_NONE, s.epi, out[v], nullptr));
        };
        if (walk) {
            // tile-walk sweep of the 16x16x32 kernel: interleaved rounds over the (gm, gn) super-block shapes
            const int nt_ = s.n / 256;
            const int cfg[][2] = {{32, 1}, {16, 2}, {16, 4}, {8, 8}, {16, nt_}, {8, nt_}, {32, nt_}, {4, nt_}};
            const int ncfg = sizeof cfg / sizeof cfg[0];
            std::vector<std::vector<float>> tt(ncfg);
            for (int r = 0; r < rounds; ++r)
                for (int c = 0; c < ncfg; ++c) {
                    g_gm = cfg[c][0]; g_gn = cfg[c][1];
                    run(1);
                    CHECK(hipEventRecord(e0));
                    for (int i = 0; i < reps; ++i) run(1);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    tt[c].push_back(ms / reps * 1e3f);
                }
            for (int c = 0; c < ncfg; ++c) {
                std::sort(tt[c].begin(), tt[c].end());
                printf("%-24s m=%d n=%d k=%d  walk gm=%2d gn=%2d: median %.1f us (%.0f TFLOP/s fp16) min %.1f us\n", s.name, s.m, s.n, s.k, cfg[c][0],
                       cfg[c][1], tt[c][tt[c].size() / 2], 2.0 * s.m * (double)s.n * 3.0 * s.k / (tt[c][tt[c].size() / 2] * 1e-6) / 1e12, tt[c][0]);
            }
            g_gm = g_gn = 0;
            fflush(stdout);
            hipFree(x); hipFree(w); hipFree(bias); hipFree(res); hipFree(at); hipFree(wt); hipFree(out[0]); hipFree(out[1]);
            continue;
        }
        run(0); run(1);
        CHECK(hipDeviceSynchronize());
        // ---- check both against float64 on sampled outputs ----
        std::vector<float> hx((size_t)s.m * s.k), hw((size_t)s.n * s.k), hb(s.n);
        CHECK(hipMemcpy(hx.data(), x, hx.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hw.data(), w, hw.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hb.data(), bias, hb.size() * 4, hipMemcpyDeviceToHost));
        double worst[2] = {0, 0}, scale = 0;
        for (int smp = 0; smp < 1024; ++smp) {
            const int m = smp < 8 ? (smp & 1 ? s.m - 1 - smp : smp) : (int)(((long long)smp * 7919 + 13) % s.m);
            const int n = smp < 8 ? (smp & 2 ? s.n - 1 - smp : smp) : (int)(((long long)smp * 104729 + 7) % s.n);
            double ref = 0.0;
            for (int k = 0; k < s.k; ++k) ref += (double)hx[(size_t)m * s.k + k] * (double)hw[(size_t)n * s.k + k];
            ref += hb[n];
            for (int v = 0; v < 2; ++v) {
                double got;
                if (!s.epi) {
                    float g, r;
                    CHECK(hipMemcpy(&g, reinterpret_cast<float *>(out[v]) + (size_t)m * s.n + n, 4, hipMemcpyDeviceToHost));
                    CHECK(hipMemcpy(&r, res + (size_t)m * s.n + n, 4, hipMemcpyDeviceToHost));
                    got = (double)g - (double)r;
                } else {
                    _Float16 ph, pl;
                    const _Float16 *Yh = reinterpret_cast<const _Float16 *>(out[v]);
                    CHECK(hipMemcpy(&ph, Yh + lemon_split::tiled_off(128, m, n, 0, s.n), 2, hipMemcpyDeviceToHost));
                    CHECK(hipMemcpy(&pl, Yh + lemon_split::tiled_off(128, m, n, 1, s.n), 2, hipMemcpyDeviceToHost));
                    got = (double)(float)ph + (double)(float)pl / 2048.0;
                }
                const double want = s.epi ? ref / (1.0 + exp(-ref)) : ref;
                worst[v] = fmax(worst[v], fabs(got - want));
                if (v == 0) scale = fmax(scale, fabs(want));
            }
        }
        // ---- interleaved timing rounds ----
        std::vector<float> t[2];
        for (int r = 0; r < rounds; ++r)
            for (int v = 0; v < 2; ++v) {
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) run(v);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                t[v].push_back(ms / reps * 1e3f);
            }
#ifdef LEMON_GEMM_PHASES
        {   // diagnostic build: where a workgroup of the 16x16x32 kernel spends its shader cycles
            if (!g_phase_dbg) CHECK(hipMalloc(&g_phase_dbg, 64));
            CHECK(hipMemset(g_phase_dbg, 0, 64));
            for (int i = 0; i < 5; ++i) run(1);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[8];
            CHECK(hipMemcpy(h, g_phase_dbg, 64, hipMemcpyDeviceToHost));
            const double wg = (double)h[3], tot = (double)(h[0] + h[1] + h[2]);
            printf("  [phases] %-22s cycles per workgroup: prologue (to the first barrier) %.0f (%.1f %%), main loop %.0f (%.1f %%), epilogue incl. store drain %.0f (%.1f %%)\n",
                   s.name, h[0] / wg, 100.0 * h[0] / tot, h[1] / wg, 100.0 * h[1] / tot, h[2] / wg, 100.0 * h[2] / tot);
            // s_memtime counts shader cycles, s_memrealtime the constant 100 MHz reference: their ratio is the clock the workgroups ran at
            printf("  [phases] %-22s mean shader clock while resident: %.0f MHz (%.0f shader cycles, %.2f us per workgroup)\n", s.name, tot / (double)h[4] * 100.0, tot / wg, h[4] / wg / 100.0);
            const double lp = (double)(h[5] + h[6] + h[7]);
            printf("  [phases] %-22s inside the loop: DMA wait + barrier B %.1f %%, fragment reads + weight block 0 + barrier B' %.1f %%, weight blocks 1-7 + DMA issue %.1f %% (%.0f cycles per k32 step; 96 MFMAs = 1536)\n",
                   s.name, 100.0 * h[5] / lp, 100.0 * h[6] / lp, 100.0 * h[7] / lp, lp / wg / (s.k / 32));
        }
#endif
        const double flop = 2.0 * s.m * (double)s.n * 3.0 * s.k;
        for (int v = 0; v < 2; ++v) {
            std::sort(t[v].begin(), t[v].end());
            const float med = t[v][t[v].size() / 2], mn = t[v][0];
            printf("%-24s m=%d n=%d k=%d  mfma %s: median %.1f us (%.0f TFLOP/s fp16) min %.1f us  max|err| %.3g of %.3g\n", s.name, s.m, s.n, s.k,
                   ln ? (v ? (ln_fold ? "16 LN-fold" : ln_emit ? "16 LN-emit" : "16 plain'") : "16 plain") : v ? "16x16x32" : "32x32x16", med, flop / (med * 1e-6) / 1e12, mn, worst[v], scale);
            if (!(worst[v] <= 2e-5 * scale + 2e-5)) { printf("  ^^^ WRONG\n"); bad = 1; }
        }
        fflush(stdout);
        hipFree(x); hipFree(w); hipFree(bias); hipFree(res); hipFree(at); hipFree(wt); hipFree(out[0]); hipFree(out[1]);
    }
    return bad;
}
