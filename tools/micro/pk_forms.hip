// pk_forms.hip -- every packed-fp32 operand form the shipped library contains, compiler-generated, beside MFMA waves.
//
// Round 4 found `v_pk_mul_f32 ... op_sel:[0,1]` (the LOW result lane reads the HIGH dword of a source pair) returning wrong low
// results in lanes 48-63 a few times per million when another wave of the SIMD issues MFMAs (profiles/r4/ln_fold_opsel_fault.txt;
// tools/micro/pk_opsel_war.hip variants 10-12).  tests/test_build_guard.py bans any packed fp32 instruction with a set op_sel bit.
// This tool asks the remaining question: are the forms that ARE in liblemon_hip.so clean?  Listed from the generated ISA of all
// eleven translation units (round 5):
//     v_pk_add_f32   plain | neg_lo/neg_hi [0,1] | neg [1,1] | op_sel_hi:[1,0] | op_sel_hi:[1,0] + neg [0,1]
//     v_pk_mul_f32   plain | op_sel_hi:[0,1] | op_sel_hi:[1,0]
//     v_pk_fma_f32   plain | op_sel_hi:[0,1,1] | [1,0,0] | [1,0,1] | [1,0,1] + neg [0,0,1] | [1,0,1] + neg [1,0,0] | [1,1,0]
// Each form below is written as vector-typed C++ that hipcc compiles to exactly that instruction (NOTHING hand-issued in the
// probe waves; `hipcc -S` of this file is checked by tools/micro/pk_forms_check.py), computed next to the same arithmetic in
// pinned scalar registers, 4 of the 8 waves of every workgroup issuing v_mfma_f32_16x16x32_f16 (two waves per SIMD, as in the
// GEMM).  Controls: the banned op_sel:[0,1] forms (multiply, add, fma) and every form again WITHOUT the MFMA waves.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_forms.hip -o .variants/pk_forms && .variants/pk_forms [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// FORM: 0..14 the shipped forms in the order above; 15..17 the banned controls (mul / add / fma with op_sel:[0,1](,0)); 18..19 the
// op_sel bit on the other operands (src0 of a multiply, src2 of an fma)
template <int FORM>
__device__ __forceinline__ void packed_and_scalar(f32x2 a, f32x2 b, f32x2 c, f32x2 &pk, float &s0, float &s1) {
    const f32x2 bx = {b[0], b[0]}, by = {b[1], b[1]}, ax = {a[0], a[0]}, cx = {c[0], c[0]};
    float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1], c0 = c[0], c1 = c[1];
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1));      // the scalar side: single registers
    switch (FORM) {
        case 0:  pk = a + b;                                   s0 = a0 + b0;                      s1 = a1 + b1; break;
        case 1:  pk = a - b;                                   s0 = a0 - b0;                      s1 = a1 - b1; break;
        case 2:  pk = -a - b;                                  s0 = -a0 - b0;                     s1 = -a1 - b1; break;
        case 3:  pk = a + bx;                                  s0 = a0 + b0;                      s1 = a1 + b0; break;
        case 4:  pk = a - bx;                                  s0 = a0 - b0;                      s1 = a1 - b0; break;
        case 5:  pk = a * b;                                   s0 = a0 * b0;                      s1 = a1 * b1; break;
        case 6:  pk = ax * b;                                  s0 = a0 * b0;                      s1 = a0 * b1; break;
        case 7:  pk = a * bx;                                  s0 = a0 * b0;                      s1 = a1 * b0; break;
        case 8:  pk = __builtin_elementwise_fma(a, b, c);      s0 = __builtin_fmaf(a0, b0, c0);   s1 = __builtin_fmaf(a1, b1, c1); break;
        case 9:  pk = __builtin_elementwise_fma(ax, b, c);     s0 = __builtin_fmaf(a0, b0, c0);   s1 = __builtin_fmaf(a0, b1, c1); break;
        case 10: pk = __builtin_elementwise_fma(a, bx, cx);    s0 = __builtin_fmaf(a0, b0, c0);   s1 = __builtin_fmaf(a1, b0, c0); break;
        case 11: pk = __builtin_elementwise_fma(a, bx, c);     s0 = __builtin_fmaf(a0, b0, c0);   s1 = __builtin_fmaf(a1, b0, c1); break;
        case 12: pk = __builtin_elementwise_fma(a, bx, -c);    s0 = __builtin_fmaf(a0, b0, -c0);  s1 = __builtin_fmaf(a1, b0, -c1); break;
        case 13: pk = __builtin_elementwise_fma(-a, bx, c);    s0 = __builtin_fmaf(-a0, b0, c0);  s1 = __builtin_fmaf(-a1, b0, c1); break;
        case 14: pk = __builtin_elementwise_fma(a, b, cx);     s0 = __builtin_fmaf(a0, b0, c0);   s1 = __builtin_fmaf(a1, b1, c0); break;
        case 15: pk = a * by;                                  s0 = a0 * b1;                      s1 = a1 * b1; break;      // banned: op_sel:[0,1]
        case 16: pk = a + by;                                  s0 = a0 + b1;                      s1 = a1 + b1; break;      // banned
        case 17: pk = __builtin_elementwise_fma(a, by, c);     s0 = __builtin_fmaf(a0, b1, c0);   s1 = __builtin_fmaf(a1, b1, c1); break;   // banned
        case 18: pk = by * a;                                  s0 = b1 * a0;                      s1 = b1 * a1; break;      // banned: op_sel:[1,0]
        default: pk = __builtin_elementwise_fma(a, c, by);     s0 = __builtin_fmaf(a0, c0, b1);   s1 = __builtin_fmaf(a1, c1, b1); break;   // banned: op_sel:[0,0,1]
    }
    asm volatile("" : "+v"(s0), "+v"(s1));
}

template <int FORM, bool MFMA>
__global__ __launch_bounds__(512, 1) void k_forms(const float *__restrict__ in, unsigned *__restrict__ bad, int iters) {
    __shared__ float2 s_b[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave >= 4) {                      // the MFMA stream of the partner wave on the SIMD (waves i and i + 4 share one)
        if (!MFMA) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        h16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
        for (int i = 0; i < iters * 4; ++i)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        if (acc[0] == 12345.678f) bad[63] = 1;      // (keeps the loop)
        return;
    }
    const float *p = in + ((size_t)blockIdx.x * 256 + tid) * 8;
    f32x2 a = {p[0], p[1]}, c = {p[2], p[3]};
    s_b[wave * 64 + lane] = make_float2(p[4], p[5]);
    unsigned n0 = 0, n1 = 0;
    for (int i = 0; i < iters; ++i) {
        // (the second operand pair arrives from LDS right in front of the arithmetic, as the row affine did in the GEMM epilogue)
        const float2 b_ = s_b[wave * 64 + ((lane + i) & 63)];
        f32x2 b = {b_.x, b_.y}, pk;
        float s0, s1;
        packed_and_scalar<FORM>(a, b, c, pk, s0, s1);
        n0 += pk[0] != s0; n1 += pk[1] != s1;
        a[0] += 0.25f; a[1] -= 0.125f;            // (the operands move: nothing is loop-invariant)
    }
    if (n0) atomicAdd(&bad[2 * (lane >> 4)], n0);
    if (n1) atomicAdd(&bad[2 * (lane >> 4) + 1], n1);
}

template <int FORM>
static void run(const float *in, unsigned *bad, int iters, const char *name) {
    for (int mf = 1; mf >= 0; --mf) {
        CHECK(hipMemset(bad, 0, 256));
        if (mf) hipLaunchKernelGGL((k_forms<FORM, true>), dim3(512), dim3(512), 0, 0, in, bad, iters);
        else    hipLaunchKernelGGL((k_forms<FORM, false>), dim3(512), dim3(512), 0, 0, in, bad, iters);
        CHECK(hipDeviceSynchronize());
        unsigned h[8];
        CHECK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
        printf("%-58s %s  mismatches (lo hi) lanes 0-15: %u %u | 16-31: %u %u | 32-47: %u %u | 48-63: %u %u   (%.3g values per cell)\n", name,
               mf ? "MFMA waves beside" : "no MFMA waves    ", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 512.0 * 4 * 16 * iters);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *in; unsigned *bad;
    const size_t n = (size_t)512 * 256 * 8;
    CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&bad, 256));
    float *h = (float *)malloc(n * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
    CHECK(hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice));
    run<0>(in, bad, iters, "0  v_pk_add_f32");
    run<1>(in, bad, iters, "1  v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]");
    run<2>(in, bad, iters, "2  v_pk_add_f32 neg_lo:[1,1] neg_hi:[1,1]");
    run<3>(in, bad, iters, "3  v_pk_add_f32 op_sel_hi:[1,0]");
    run<4>(in, bad, iters, "4  v_pk_add_f32 op_sel_hi:[1,0] neg [0,1]");
    run<5>(in, bad, iters, "5  v_pk_mul_f32");
    run<6>(in, bad, iters, "6  v_pk_mul_f32 op_sel_hi:[0,1]");
    run<7>(in, bad, iters, "7  v_pk_mul_f32 op_sel_hi:[1,0]");
    run<8>(in, bad, iters, "8  v_pk_fma_f32");
    run<9>(in, bad, iters, "9  v_pk_fma_f32 op_sel_hi:[0,1,1]");
    run<10>(in, bad, iters, "10 v_pk_fma_f32 op_sel_hi:[1,0,0]");
    run<11>(in, bad, iters, "11 v_pk_fma_f32 op_sel_hi:[1,0,1]");
    run<12>(in, bad, iters, "12 v_pk_fma_f32 op_sel_hi:[1,0,1] neg [0,0,1]");
    run<13>(in, bad, iters, "13 v_pk_fma_f32 op_sel_hi:[1,0,1] neg [1,0,0]");
    run<14>(in, bad, iters, "14 v_pk_fma_f32 op_sel_hi:[1,1,0]");
    run<15>(in, bad, iters, "15 BANNED v_pk_mul_f32 op_sel:[0,1]");
    run<16>(in, bad, iters, "16 BANNED v_pk_add_f32 op_sel:[0,1]");
    run<17>(in, bad, iters, "17 BANNED v_pk_fma_f32 op_sel:[0,1,0]");
    run<18>(in, bad, iters, "18 BANNED v_pk_mul_f32 op_sel on src0");
    run<19>(in, bad, iters, "19 BANNED v_pk_fma_f32 op_sel on src2");
    return 0;
}
