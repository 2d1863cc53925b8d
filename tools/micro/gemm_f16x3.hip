// gemm_f16x3.hip -- go/no-go for a hand-written split-fp16 GEMM (the lemon_linear_f16x3 arithmetic: hi.hi + hi.lo + lo.hi,
// fp32 accumulate) against the library kernels the towers use today (hipBLASLt HSS, ~1.05 PFLOP/s on random operands at
// m = 50 000, k' = 2 304).  What a kernel that KNOWS the operand structure can save: the activation's hi part is staged once
// (the library's [hi | hi | lo] row stages it twice), operands are pre-packed tile-major in MFMA fragment order (contiguous
// LDS-DMA, linear conflict-free ds_read_b128, no swizzle), and the epilogue is ours to fuse.
//
// Workgroup: 256 x 256 output tile, 4 waves of 128 x 128 (16 accumulator tiles = 256 AccVGPRs), one original-k16 step per
// stage: A_hi, A_lo (2 x 8 KB) + W_hi, W_lo, W_his (3 x 8 KB) = 40 KB, ring of NB stages, 48 MFMAs per wave per barrier,
// 20 fragment reads per 48 MFMAs (427 B of LDS per MFMA).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gemm_f16x3.hip -o .variants/gemm_f16x3 && .variants/gemm_f16x3 [m n k]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef TMV
#define TMV 256                            // rows of the activation tile: 256 (one workgroup per CU, 256 AccVGPRs) or 128 (two per CU)
#endif
constexpr int TM = TMV, TN = 256;
constexpr int IB = TM / 64;                // 32-row activation blocks per wave
constexpr int BLK = 8192;                  // one segment of one k16 step of a 256-row weight tile: 8 row blocks x 1 KB
constexpr int BLKA = TM * 32;              // ... of an activation tile
#ifndef WSEG
#define WSEG 2                             // 2: W_his = W_hi * 2^-11 is made in registers (v_pk_mul_f16, exact); 3: staged like the rest
#endif
constexpr int STAGE = 2 * BLKA + WSEG * BLK;    // A_hi A_lo W_hi W_lo [W_his]
#ifndef NB
#define NB 4
#endif
constexpr int LA = NB - 1;
#ifndef EPI
#define EPI 0                              // 0: y = alpha acc + bias, fp32 row-major; 1: silu(alpha acc + bias) split to [hi | lo 2^11] fp16, tile-major
#endif
#ifndef ABL
#define ABL 0                              // timing ablations (results wrong): 1 no DMA after the prologue, 2 no fragment reads, 4 no barriers
#endif
#ifndef GM
#define GM 16
#endif
#ifndef GN
#define GN 2
#endif

// tile-major, fragment-linear operand layout: [tile][k16 step][segment][row block of 32][k half][row in block][8 k]
__host__ __device__ inline size_t pack_off(int trows, int nseg, int ks_total, int row, int k, int seg) {
    const int tile = row / trows, r = row % trows, rb = r / 32, rr = r % 32, ks = k / 16, kh = (k % 16) / 8, e = k % 8;
    return ((((size_t)tile * ks_total + ks) * nseg + seg) * (trows * 32)) / 2 + (size_t)rb * 512 + kh * 256 + rr * 8 + e;    // in halves
}

__global__ void k_fill(_Float16 *g, size_t n, float scale, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned r = ((unsigned)i + seed) * 2654435761u; r ^= r >> 15; r *= 2246822519u; r ^= r >> 13; r *= 3266489917u; r ^= r >> 16;
        // sum of four uniforms: roughly normal, like the library probe's operands
        const float u = ((r & 255) + ((r >> 8) & 255) + ((r >> 16) & 255) + (r >> 24)) * (1.0f / 255.0f) - 2.0f;
        g[i] = (_Float16)(u * scale);
    }
}

__device__ __forceinline__ void mfma(f32x16 &acc, const h16x8 &a, const h16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void dma1k(const char *src, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src), "s"(lds) : "memory");
}
__device__ __forceinline__ h16x8 lds128(unsigned addr, int off) {
    h16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}

// Y[m][n] = alpha * sum_k (a_hi w_hi + a_hi w_lo + a_lo' w_his) + bias[n]
__global__ __launch_bounds__(256, TM == 128 ? 2 : 1) void k_gemm(const char *__restrict__ At, const char *__restrict__ Wt, const float *__restrict__ bias,
                                                 float *__restrict__ Y, int M, int N, int KS, int m_tiles, int n_tiles, float alpha) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs.  An XCD works through super-blocks of
    // GM x GN tiles (32 = its CUs): the GN weight tile columns (1.2 MB each) stay in its 4 MB L2 and are shared by GM
    // workgroups walking k in step, the activation tile rows stream.
    int mt, nt;
    {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int gm_n = (m_tiles + GM - 1) / GM, gn_n = (n_tiles + GN - 1) / GN;
        const int blk = (q / (GM * GN)) * 8 + xcd, pos = q % (GM * GN);
        if (blk >= gm_n * gn_n) return;
        const int g = blk / gn_n, pn = blk % gn_n;
        mt = g * GM + pos / GN; nt = pn * GN + pos % GN;
        if (mt >= m_tiles || nt >= n_tiles) return;
    }
    const char *a_src = At + (size_t)mt * KS * 2 * BLKA;
    const char *w_src = Wt + (size_t)nt * KS * 3 * BLK;      // (the packed operand keeps its third segment either way)
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)smem;
    // DMA shares: wave w moves A bytes [4 KB w, +4 KB) and W bytes [6 KB w, +6 KB) of a stage: ten 1-KB instructions
    const unsigned va = (unsigned)(wave * (BLKA / 2) + lane * 16), vw = (unsigned)(wave * (WSEG * 2048) + lane * 16);
    auto issue = [&](int ks, int slot) {
        if ((ABL & 1) && ks >= LA) return;
        const char *as = a_src + (size_t)ks * 2 * BLKA, *ws = w_src + (size_t)ks * 3 * BLK;
        const unsigned la = lds0 + slot * STAGE + wave * (BLKA / 2), lw = lds0 + slot * STAGE + 2 * BLKA + wave * (WSEG * 2048);
#pragma unroll
        for (int j = 0; j < BLKA / 2048; ++j) dma1k(as, va + j * 1024, la + j * 1024);
#pragma unroll
        for (int j = 0; j < 2 * WSEG; ++j) dma1k(ws, vw + j * 1024, lw + j * 1024);
    };
    f32x16 acc[4][IB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < IB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < LA; ++s)
        if (s < KS) issue(s, s);
    // fragment addresses inside a stage: A block (4 wm + i) of segment s, W block (4 wn + j) of segment s
    const unsigned fa = lds0 + wm * (IB * 1024) + lane * 16, fw = lds0 + 2 * BLKA + wn * 4096 + lane * 16;
    h16x8 af[2][IB][2], wf[2][4][3];
#define READ_FRAGS(set, slot)                                                                                   \
    do {                                                                                                        \
        if ((ABL & 2) && ((set) != 0 || (slot) != 0)) break;                                                    \
        const unsigned pa_ = fa + (slot) * STAGE, pw_ = fw + (slot) * STAGE;                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) {                                                     \
            af[set][i_][0] = lds128(pa_, i_ * 1024); af[set][i_][1] = lds128(pa_, BLKA + i_ * 1024);            \
        }                                                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
            wf[set][j_][0] = lds128(pw_, j_ * 1024); wf[set][j_][1] = lds128(pw_, BLK + j_ * 1024);             \
            if (WSEG == 3) wf[set][j_][2] = lds128(pw_, 2 * BLK + j_ * 1024);                                   \
        }                                                                                                       \
    } while (0)
#define DO_MFMAS(set)                                                                                           \
    do {                                                                                                        \
        if (WSEG == 2) { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) wf[set][j_][2] = wf[set][j_][0] * (_Float16)0.00048828125f; } \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) mfma(acc[j_][i_], wf[set][j_][0], af[set][i_][0]); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) mfma(acc[j_][i_], wf[set][j_][1], af[set][i_][0]); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) mfma(acc[j_][i_], wf[set][j_][2], af[set][i_][1]); \
    } while (0)
    // stage 0 landed?  (LA stages in flight, 4 + 2 WSEG instructions each)
#define WAITN_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAITN(n) WAITN_(n)
#define WAITV(n) do { if ((n) == 20) WAITN(20); else if ((n) == 16) WAITN(16); else if ((n) == 12) WAITN(12); else if ((n) == 10) WAITN(10); else if ((n) == 8) WAITN(8); else if ((n) == 6) WAITN(6); else WAITN(0); } while (0)
#define WAIT_STAGE() WAITV((LA - 1) * (BLKA / 2048 + 2 * WSEG))
    WAIT_STAGE();
    __syncthreads();
    READ_FRAGS(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define PIN_ACC() do { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) asm volatile("" : "+a"(acc[j_][i_])); } while (0)
    PIN_ACC();
    for (int t = 0; t < KS; t += 2) {
        PIN_ACC();          // (the loop-carried accumulators stay AccVGPRs: without it hipcc homes them in VGPRs and copies 256 registers per step)
        // ---- step t (fragment set 0); bring in stage t+1's fragments (set 1) under its MFMAs ----
        if (t + LA < KS) issue(t + LA, (t + LA) % NB);       // slot of stage t-1: everyone is past the barrier behind its reads
        if (t + 1 < KS) {
            if (t + LA < KS) { WAIT_STAGE(); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(ABL & 4)) __syncthreads();
            READ_FRAGS(1, (t + 1) % NB);
        }
        DO_MFMAS(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 1 >= KS) break;
        // ---- step t+1 (set 1) ----
        if (t + 1 + LA < KS) issue(t + 1 + LA, (t + 1 + LA) % NB);
        if (t + 2 < KS) {
            if (t + 1 + LA < KS) { WAIT_STAGE(); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(ABL & 4)) __syncthreads();
            READ_FRAGS(0, (t + 2) % NB);
        }
        DO_MFMAS(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- epilogue: lane holds m = l%32 of A block i, n = 8 (e/4) + 4 (l/32) + e%4 of W block j ----
    const int l31 = lane & 31, h = lane >> 5;
    if ((ABL & 8) && acc[0][0][0] != 123.456f) return;          // timing ablation: no epilogue stores
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int m = mt * TM + wm * (IB * 32) + i * 32 + l31;
        if (EPI == 0 && m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = nt * TN + wn * 128 + j * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + 8 * g;
                if (n >= N) continue;
                const float4 b = *reinterpret_cast<const float4 *>(bias + n);
                float o[4] = {alpha * acc[j][i][4 * g] + b.x, alpha * acc[j][i][4 * g + 1] + b.y,
                              alpha * acc[j][i][4 * g + 2] + b.z, alpha * acc[j][i][4 * g + 3] + b.w};
                if (EPI == 0) {
                    *reinterpret_cast<float4 *>(Y + (size_t)m * N + n) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    // the NEXT GEMM's activation operand (its k = this n), tile-major like At: row block (m/32)%(TM/32) of tile
                    // m/TM, k16 step n/16, k half (n%16)/8, 16-byte slot m%32, bytes 2 (n%8): lanes l and l+32 fill one slot,
                    // a wave's store instruction covers 512 contiguous bytes
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = o[e] / (1.0f + __expf(-o[e]));             // SiLU
                        const _Float16 vh = (_Float16)v;
                        hi[e] = vh; lo[e] = (_Float16)((v - (float)vh) * 2048.0f);
                    }
                    const size_t KS2 = (size_t)N / 16;
                    char *base = reinterpret_cast<char *>(Y) + (((size_t)(m / TM) * KS2 + n / 16) * 2) * BLKA
                               + ((m % TM) / 32) * 1024 + ((n % 16) / 8) * 512 + (m % 32) * 16 + (n % 8) * 2;
                    *reinterpret_cast<h4 *>(base) = hi;
                    *reinterpret_cast<h4 *>(base + BLKA) = lo;
                }
            }
        }
    }
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 50000, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int mt = (M + TM - 1) / TM, nt = (N + TN - 1) / TN, KS = K / 16;
    if (K % 16 || N % 4) { fprintf(stderr, "k %% 16, n %% 4\n"); return 1; }
    const size_t a_halves = (size_t)mt * KS * 2 * BLKA / 2, w_halves = (size_t)nt * KS * 3 * BLK / 2;
    _Float16 *At, *Wt; float *bias, *Y;
    CHECK(hipMalloc(&At, a_halves * 2)); CHECK(hipMalloc(&Wt, w_halves * 2));
    CHECK(hipMalloc(&bias, (size_t)nt * TN * 4)); CHECK(hipMalloc(&Y, (size_t)mt * TM * N * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, At, a_halves, 1.0f, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Wt, w_halves, 1.0f, 77u);
    CHECK(hipMemset(bias, 0, (size_t)nt * TN * 4));
    CHECK(hipDeviceSynchronize());
    const int total = mt * nt;
    const int blocks = ((mt + GM - 1) / GM) * ((nt + GN - 1) / GN);
    const int grid = ((blocks + 7) / 8) * GM * GN * 8;
    const size_t lds = (size_t)NB * STAGE;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto run = [&]() {
        hipLaunchKernelGGL(k_gemm, dim3(grid), dim3(256), lds, 0, reinterpret_cast<const char *>(At), reinterpret_cast<const char *>(Wt),
                           bias, Y, M, N, KS, mt, nt, 1.0f);
    };
    run();
    CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
    // ---- check 256 sampled outputs against a double-precision sum over the same packed operands ----
    std::vector<_Float16> ha(a_halves), hw(w_halves);
    CHECK(hipMemcpy(ha.data(), At, a_halves * 2, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hw.data(), Wt, w_halves * 2, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (int s = 0; s < 256; ++s) {
        const int m = (int)(((long long)s * 7919 + 13) % M), n = (int)(((long long)s * 104729 + 7) % N);
        double ref = 0.0;
        for (int k = 0; k < K; ++k) {
            const double ah = (double)(float)ha[pack_off(TM, 2, KS, m, k, 0)], al = (double)(float)ha[pack_off(TM, 2, KS, m, k, 1)];
            const double wh = (double)(float)hw[pack_off(TN, 3, KS, n, k, 0)], wl = (double)(float)hw[pack_off(TN, 3, KS, n, k, 1)],
                         ws = WSEG == 2 ? (double)(float)(_Float16)((float)hw[pack_off(TN, 3, KS, n, k, 0)] * 0.00048828125f)
                                        : (double)(float)hw[pack_off(TN, 3, KS, n, k, 2)];
            ref += ah * wh + ah * wl + al * ws;
        }
        float got;
        if (EPI == 0) CHECK(hipMemcpy(&got, Y + (size_t)m * N + n, 4, hipMemcpyDeviceToHost));
        else {
            _Float16 ph, pl;
            const _Float16 *Yh = reinterpret_cast<const _Float16 *>(Y);
            CHECK(hipMemcpy(&ph, Yh + pack_off(TM, 2, N / 16, m, n, 0), 2, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&pl, Yh + pack_off(TM, 2, N / 16, m, n, 1), 2, hipMemcpyDeviceToHost));
            got = (float)ph + (float)pl / 2048.0f;
            ref = ref / (1.0 + exp(-ref));
        }
        worst = fmax(worst, fabs((double)got - ref)); scale = fmax(scale, fabs(ref));
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) run();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / reps);
    }
    printf("gemm_f16x3 EPI=%d TM=%d NB=%d GM=%d GN=%d WS=%d ABL=%d m=%d n=%d k=%d (k'=%d): %.1f us  %.1f TFLOP/s fp16 (%.1f TFLOP/s fp32-equivalent)  max |err| %.3g of %.3g  tiles %d\n",
           EPI, TM, NB, GM, GN, WSEG, ABL, M, N, K, 3 * K, best * 1e3, 2.0 * M * N * 3.0 * K / (best * 1e-3) / 1e12, 2.0 * M * N * (double)K / (best * 1e-3) / 1e12, worst, scale, total);
    return worst <= 1e-3 * scale + 1e-3 ? 0 : 2;
}
