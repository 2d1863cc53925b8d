// mfma_peak.hip -- what the matrix pipe of one MI355X sustains with NO memory traffic at all: every wave issues
// back-to-back MFMAs on registers (4 or 8 independent accumulator tiles), 1..3 waves per SIMD.  The number DESIGN.md
// quotes next to the data-sheet peak when it prices k_scan_f32 / k_scan_bf16_qs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o .variants/mfma_peak && .variants/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ACC>
__global__ __launch_bounds__(256) void k_f32(float *out, int iters, float a, float b) {
    f32x16 acc[ACC];
    for (int i = 0; i < ACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / ACC * 4; ++r)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < ACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
}

template <int ACC>
__global__ __launch_bounds__(256) void k_bf16(float *out, int iters, float a) {
    f32x16 acc[ACC];
    for (int i = 0; i < ACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    bf16x8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (__bf16)(a + e); bv[e] = (__bf16)(a - e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / ACC * 4; ++r)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < ACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
}

// bf16 MFMAs on RANDOM operands (8 rotating A fragments, 4 B fragments): what the matrix pipe sustains when its inputs
// toggle like real data -- the board's power management lowers the shader clock (reported next to the rate)
__global__ __launch_bounds__(256) void k_bf16_random(float *out, int iters, unsigned seed, int zero) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    bf16x8 av[8], bv[4];
    unsigned r = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) {
        r ^= r << 13; r ^= r >> 17; r ^= r << 5;
        av[i][e] = zero ? (__bf16)0.0f : (__bf16)(((int)(r & 1023) - 512) * (1.0f / 4096.0f));
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) {
        r ^= r << 13; r ^= r >> 17; r ^= r << 5;
        bv[i][e] = zero ? (__bf16)0.0f : (__bf16)(((int)(r & 1023) - 512) * (1.0f / 4096.0f));
    }
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(rr + i) & 7], bv[(rr >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<unsigned long long *>(out)[1] = __builtin_amdgcn_s_memtime() - clk0;
        reinterpret_cast<unsigned long long *>(out)[2] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

// the same with v_mfma_f32_16x16x32_bf16 (a quarter of the accumulator registers per instruction, half per flop)
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_bf16_random_16(float *out, int iters, unsigned seed, int zero) {
    f32x4v acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.0f;
    bf16x8 av[8], bv[4];
    unsigned r = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) {
        r ^= r << 13; r ^= r >> 17; r ^= r << 5;
        av[i][e] = zero ? (__bf16)0.0f : (__bf16)(((int)(r & 1023) - 512) * (1.0f / 4096.0f));
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) {
        r ^= r << 13; r ^= r >> 17; r ^= r << 5;
        bv[i][e] = zero ? (__bf16)0.0f : (__bf16)(((int)(r & 1023) - 512) * (1.0f / 4096.0f));
    }
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[(rr + i) & 7], bv[(rr >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<unsigned long long *>(out)[1] = __builtin_amdgcn_s_memtime() - clk0;
        reinterpret_cast<unsigned long long *>(out)[2] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *out; CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(out, 0, 64));
    const int iters = 4000;                       // x 64 MFMAs per wave
    for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {
        const int grid = cus * wg_per_cu;
        const double mf = (double)grid * 4 * iters * 64;     // MFMAs in the launch
        double ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
        printf("f32 32x32x2   4 acc tiles  %d wave(s)/SIMD  %8.3f ms  %7.1f TFLOP/s\n", wg_per_cu, ms, mf * 4096 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
        printf("f32 32x32x2   8 acc tiles  %d wave(s)/SIMD  %8.3f ms  %7.1f TFLOP/s\n", wg_per_cu, ms, mf * 4096 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_bf16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
        printf("bf16 32x32x16 4 acc tiles  %d wave(s)/SIMD  %8.3f ms  %7.1f TFLOP/s\n", wg_per_cu, ms, mf * 32768 / ms / 1e9);
    }
    for (int zero = 1; zero >= 0; --zero) {
        const int it2 = 60000;                    // ~100 ms: long enough for the power management to settle
        const double mf = (double)cus * 4 * it2 * 64;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_bf16_random, dim3(cus), dim3(256), 0, 0, out, it2, 12345u, zero); });
        unsigned long long h[3];
        CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
        printf("bf16 32x32x16 sustained, %s operands, 1 wave/SIMD  %8.3f ms  %7.1f TFLOP/s  shader clock %4.0f MHz\n",
               zero ? "all-zero" : "random  ", ms, mf * 32768 / ms / 1e9, h[2] ? 100.0 * (double)h[1] / (double)h[2] : 0.0);
    }
    for (int zero = 1; zero >= 0; --zero) {
        const int it2 = 60000;
        const double mf = (double)cus * 4 * it2 * 128;       // 16x16x32 MFMAs: 16 384 flop each
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_bf16_random_16, dim3(cus), dim3(256), 0, 0, out, it2, 12345u, zero); });
        unsigned long long h[3];
        CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
        printf("bf16 16x16x32 sustained, %s operands, 1 wave/SIMD  %8.3f ms  %7.1f TFLOP/s  shader clock %4.0f MHz\n",
               zero ? "all-zero" : "random  ", ms, mf * 16384 / ms / 1e9, h[2] ? 100.0 * (double)h[1] / (double)h[2] : 0.0);
    }
    return 0;
}
