// qs2_loop.hip -- go/no-go for a Q-stationary bf16 scan loop with TWO 32-query blocks per wave (d = 768).
//
// k_scan_bf16_qs keeps one 32-query block per wave in registers (192 AccVGPRs) and reads the whole 128-row database tile
// from LDS for it: 1 KB of LDS per MFMA, one 128-row x 64-k slice (16 KB) delivered per 16 MFMAs of a wave.  Here a wave
// owns 64 queries: block 0 in the 192 AccVGPRs next to the 64 accumulator registers, block 1 in architectural VGPRs
// except its last PARK k-steps, which live in LDS (lane-linear, conflict-free).  A database tile is 64 rows; every
// fragment read feeds two MFMAs, so LDS bytes per MFMA and global -> LDS bytes per flop both HALVE (a workgroup now covers
// 256 queries) at the same 32 MFMAs per wave between barriers.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/qs2_loop.hip -o .variants/qs2_loop && .variants/qs2_loop [tiles] [random]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int DW = 384;                    // row pitch in 4-byte words (768 bf16)
constexpr int KT2 = 6;                     // stages per tile (2 x 64-wide k-slices each)
constexpr int RT = 64;                     // database rows per tile
constexpr int STG = 2 * RT * 32;           // floats per stage (16 KB)

__device__ __forceinline__ int swz(int r, int c) { return r * 32 + 4 * (c ^ ((r >> 1) & 7)); }
__device__ __forceinline__ void mfma_a(f32x16 &acc, bf16x8 a, const bf16x8 &bq) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(bq));
}
__device__ __forceinline__ void mfma_v(f32x16 &acc, bf16x8 a, const bf16x8 &bq) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(bq));
}

__global__ void k_fill(unsigned *g, size_t n, int random) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned r = (unsigned)i * 2654435761u; r ^= r >> 15; r *= 2246822519u; r ^= r >> 13;
        g[i] = random ? (((r & 0x807fu) | 0x3d00u) | ((((r >> 16) & 0x807fu) | 0x3d00u) << 16)) : 0u;
    }
}

// PARK: k-steps of query block 1 whose fragments are read from LDS instead of registers; DEEP: fragment reads two
// k-steps ahead (three sets) instead of one
template <int PARK, bool DEEP>
__global__ __launch_bounds__(256, 1) void k_loop2(const float *__restrict__ g, float *out, int tiles, unsigned spread, int lockstep) {
    __shared__ __attribute__((aligned(16))) float smem[4 * STG + 4 * (PARK ? PARK : 1) * 256];
    float *s_x = smem;                       // 64 KB ring
    float *s_q = smem + 4 * STG;             // [4 waves][PARK][64 lanes x 16 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 4 * STG; i += 256) s_x[i] = 0.0f;
    for (int i = tid; i < 4 * PARK * 256; i += 256) s_q[i] = 0.001f * (float)(i % 97 - 48);
    __syncthreads();
    constexpr int NR = 48 - PARK;            // block-1 fragments kept in registers
    bf16x8 q0[48], q1[NR];
#pragma unroll
    for (int s = 0; s < 48; ++s) for (int e = 0; e < 8; ++e) q0[s][e] = (__bf16)(0.01f * (float)(((s * 8 + e) * 37 + lane * 11) % 23 - 11));
#pragma unroll
    for (int s = 0; s < NR; ++s) for (int e = 0; e < 8; ++e) q1[s][e] = (__bf16)(0.01f * (float)(((s * 8 + e) * 29 + lane * 13) % 19 - 9));
    f32x16 acc00, acc01, acc10, acc11;
    for (int e = 0; e < 16; ++e) { acc00[e] = 0; acc01[e] = 0; acc10[e] = 0; acc11[e] = 0; }
    unsigned frag_addr[4];
    for (int u = 0; u < 4; ++u)
        frag_addr[u] = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_x + swz(l31, 2 * u + h));
    const unsigned vq = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)(s_q + (wave * PARK) * 256 + 4 * lane);
    const int t0 = lockstep ? 0 : (int)((blockIdx.x * 61u) % spread);
    const bf16x8 z8 = {};
    bf16x8 fa0 = z8, fa1 = z8, fb0 = z8, fb1 = z8, fc0 = z8, fc1 = z8, qa = z8, qb = z8, qc = z8;
    // stage s: tile (t0 + s / KT2) % spread, k-slices 2 (s % KT2), 2 (s % KT2) + 1; a wave moves rows 16 wave + 8 i + lane/8
#define SRC(s_, sb_) (g + (size_t)((t0 + (s_) / KT2) % spread) * RT * DW + (2 * ((s_) % KT2) + (sb_)) * 32)
#define DMA1(src_, i_, dst_)                                                                           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((src_) + (size_t)(16 * wave + 8 * (i_) + (lane >> 3)) * DW + \
                                         4 * ((lane & 7) ^ (((8 * (i_) + (lane >> 3)) >> 1) & 7))),        \
                                     (__attribute__((address_space(3))) void *)((dst_) + (16 * wave + 8 * (i_)) * 32), 16, 0, 0)
#define DMA_STAGE(s_)                                                                                  \
    do {                                                                                               \
        float *d0_ = s_x + ((s_) & 3) * STG;                                                           \
        DMA1(SRC(s_, 0), 0, d0_); DMA1(SRC(s_, 0), 1, d0_);                                            \
        DMA1(SRC(s_, 1), 0, d0_ + RT * 32); DMA1(SRC(s_, 1), 1, d0_ + RT * 32);                        \
    } while (0)
    // fragment set S for k-step KSV of the stage at LDS byte offset SB: rows l31 and l31 + 32 (+ the parked query fragment)
#define LOADS(S, KSV, SB)                                                                              \
    do {                                                                                               \
        const unsigned va_ = frag_addr[(KSV) & 3] + (SB) + (((KSV) & 4) ? (unsigned)(RT * 128) : 0u);   \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"                         \
                     : "=&v"(f##S##0), "=&v"(f##S##1) : "v"(va_) : "memory");                          \
        if (ks0 + (KSV) >= NR) {                                                                       \
            const unsigned vp_ = vq + (unsigned)((ks0 + (KSV) - NR) * 1024);                           \
            asm volatile("ds_read_b128 %0, %1" : "=&v"(q##S) : "v"(vp_) : "memory");                   \
        }                                                                                              \
    } while (0)
#define NRD(KSV) ((KSV) > 7 ? 0 : (ks0 + (KSV) >= NR ? 3 : 2))
#define WAITN(S, N)                                                                                    \
    do {                                                                                               \
        switch (N) {                                                                                   \
            case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(q##S)); break; \
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(q##S)); break; \
            case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(q##S)); break; \
            case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(q##S)); break; \
            case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(q##S)); break; \
            default: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f##S##0), "+v"(f##S##1), "+v"(q##S)); break; \
        }                                                                                              \
    } while (0)
#define STEP(S, KSV)                                                                                   \
    do {                                                                                               \
        const int ks_ = ks0 + (KSV);                                                                   \
        mfma_a(acc00, f##S##0, q0[ks_]); mfma_a(acc01, f##S##1, q0[ks_]);                              \
        if (ks_ < NR) { mfma_v(acc10, f##S##0, q1[ks_ < NR ? ks_ : 0]); mfma_v(acc11, f##S##1, q1[ks_ < NR ? ks_ : 0]); } \
        else          { mfma_v(acc10, f##S##0, q##S); mfma_v(acc11, f##S##1, q##S); }                  \
    } while (0)
    const int total = tiles * KT2;
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    DMA_STAGE(0); DMA_STAGE(1); DMA_STAGE(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int jl = 0; jl < tiles; ++jl) {
#pragma clang loop unroll(full)
        for (int kt = 0; kt < KT2; ++kt) {
            const int t = jl * KT2 + kt;
            DMA_STAGE(t + 3);
            const unsigned sb = (unsigned)((t & 3) * STG * 4);
            const int ks0 = 8 * kt;
            if (!DEEP) {
                LOADS(a, 0, sb);
                LOADS(b, 1, sb); WAITN(a, NRD(1)); STEP(a, 0);
                LOADS(a, 2, sb); WAITN(b, NRD(2)); STEP(b, 1);
                LOADS(b, 3, sb); WAITN(a, NRD(3)); STEP(a, 2);
                LOADS(a, 4, sb); WAITN(b, NRD(4)); STEP(b, 3);
                LOADS(b, 5, sb); WAITN(a, NRD(5)); STEP(a, 4);
                LOADS(a, 6, sb); WAITN(b, NRD(6)); STEP(b, 5);
                LOADS(b, 7, sb); WAITN(a, NRD(7)); STEP(a, 6);
                                 WAITN(b, 0);      STEP(b, 7);
            } else {
                LOADS(a, 0, sb); LOADS(b, 1, sb);
                LOADS(c, 2, sb); WAITN(a, NRD(1) + NRD(2)); STEP(a, 0);
                LOADS(a, 3, sb); WAITN(b, NRD(2) + NRD(3)); STEP(b, 1);
                LOADS(b, 4, sb); WAITN(c, NRD(3) + NRD(4)); STEP(c, 2);
                LOADS(c, 5, sb); WAITN(a, NRD(4) + NRD(5)); STEP(a, 3);
                LOADS(a, 6, sb); WAITN(b, NRD(5) + NRD(6)); STEP(b, 4);
                LOADS(b, 7, sb); WAITN(c, NRD(6) + NRD(7)); STEP(c, 5);
                                 WAITN(a, NRD(7));          STEP(a, 6);
                                 WAITN(b, 0);               STEP(b, 7);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // 2 younger stages x 4 DMAs may still be in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" : "+a"(acc00), "+a"(acc01), "+a"(acc10), "+a"(acc11));
    float s = 0.0f;
    for (int e = 0; e < 16; ++e) s += acc00[e] + acc01[e] + acc10[e] + acc11[e];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && tid == 0) {
        reinterpret_cast<unsigned long long *>(out)[1] = __builtin_amdgcn_s_memtime() - clk0;
        reinterpret_cast<unsigned long long *>(out)[2] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

static int g_tiles = 2400;
template <int PARK, bool DEEP>
static void run(const float *g, float *out, int cus, const char *what, unsigned spread, int lockstep) {
    const int tiles = g_tiles, grid = cus;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_loop2<PARK, DEEP>), dim3(grid), dim3(256), 0, 0, g, out, tiles, spread, lockstep);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_loop2<PARK, DEEP>), dim3(grid), dim3(256), 0, 0, g, out, tiles, spread, lockstep);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double flop = (double)grid * 4 * tiles * KT2 * 32 * 32768.0;
    unsigned long long h[3];
    CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
    printf("qs2 park %2d %s  %-58s %7.3f ms  %7.1f TFLOP/s  %.3f of 2500  shader clock %4.0f MHz\n", PARK, DEEP ? "two ahead" : "one ahead",
           what, best, flop / best / 1e9, flop / best / 1e9 / 2500.0, h[2] ? 100.0 * (double)h[1] / (double)h[2] : 0.0);
}

int main(int argc, char **argv) {
    if (argc > 1) g_tiles = atoi(argv[1]);      // 64-row tiles: 2400 = the work of 1200 tiles of qs_loop
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *g, *out;
    const size_t rows = 304 * 128;
    CHECK(hipMalloc(&g, rows * DW * 4));
    const int random = argc > 2 ? atoi(argv[2]) : 1;
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, reinterpret_cast<unsigned *>(g), rows * DW, random);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMalloc(&out, 64));
    run<16, false>(g, out, cus, "all workgroups walk the same 600 tiles in step", 600, 1);
    run<16, true>(g, out, cus, "all workgroups walk the same 600 tiles in step", 600, 1);
    run<12, false>(g, out, cus, "all workgroups walk the same 600 tiles in step", 600, 1);
    run<12, true>(g, out, cus, "all workgroups walk the same 600 tiles in step", 600, 1);
    run<24, false>(g, out, cus, "all workgroups walk the same 600 tiles in step", 600, 1);
    run<16, false>(g, out, cus, "every workgroup on the same 8 tiles", 8, 0);
    run<16, false>(g, out, cus, "workgroups spread over 600 tiles (Infinity Cache)", 600, 0);
    return 0;
}
