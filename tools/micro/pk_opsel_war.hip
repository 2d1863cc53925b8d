// Reproducer attempt for the packed-fp32 fault of the folded-LayerNorm epilogue (round 4, DESIGN.md "hardware facts").
// The failing compiler-generated sequence was
//     v_pk_mul_f32 T, C.xy, AFF op_sel:[0,1]            ; t.xy = c.xy * aff.y   (low lane reads the HIGH dword of AFF)
//     v_pk_add_f32 ...
//     v_pk_fma_f32 O.xy, V.xy, AFF, T op_sel_hi:[1,0,1] ; o.xy = v.xy * aff.x + t.xy
//     v_pk_mul_f32 T, C.zw, AFF op_sel:[0,1]            ; t.zw -> the SAME register pair T, right behind its last reader
//     v_pk_fma_f32 O.zw, V.zw, AFF, T op_sel_hi:[1,0,1]
// with wrong o.x / o.z in lanes 48-63 now and then, by exactly (t.z - t.x) aff... i.e. as if the second multiply's low result had
// landed in T before the first fma read it.  Here: the same five instructions as asm in a loop, half of the waves of every
// workgroup issuing MFMAs next to them (two waves per SIMD, as in the GEMM), results checked against scalar arithmetic.
// Prints the mismatch count per 16-lane group and output component for seven variants of the sequence.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// VARIANT 0: the sequence (fma reads T, the next instruction -- a packed multiply -- writes T); 1: the same without the MFMA
// waves; 2: s_nop 0 between the fma and the multiply that overwrites T; 3: the second multiply writes another pair (no WAR);
// 4: T overwritten by two scalar v_mul_f32 instead of the packed one; 5: an independent packed add between the fma and the multiply;
// 6: as 0 but the first multiply without op_sel (aff.y broadcast into its own pair)
template <int VARIANT>
__global__ __launch_bounds__(512, 1) void k_probe(const float *__restrict__ in, unsigned *__restrict__ bad, int iters) {
    __shared__ float2 s_aff[64 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave >= 4) {                      // the MFMA stream of the partner wave on the SIMD (waves i and i + 4 share one)
        if (VARIANT == 1) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        h16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
        for (int i = 0; i < iters * 6; ++i)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        if (acc[0] == 12345.678f) bad[31] = 1;      // (keeps the loop)
        return;
    }
    const float *p = in + ((size_t)blockIdx.x * 512 + tid) * 12;
    f32x2 cxy = {p[0], p[1]}, czw = {p[2], p[3]}, vxy = {p[4], p[5]}, vzw = {p[6], p[7]};
    s_aff[wave * 64 + lane] = make_float2(p[8], p[9]);
    unsigned nb[4] = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        f32x2 aff, t, t2, oxy, ozw, filler = {p[10], p[11]};
        // the pair arrives from LDS right in front of the sequence, as in the kernel
        if (VARIANT == 7) { const float2 a_ = s_aff[wave * 64 + ((lane + i) & 63)]; aff[0] = a_.x; aff[1] = a_.y; asm volatile("" : "+v"(aff)); }   // compiler's own LDS read + wait
        else if (VARIANT == 8) { aff[0] = p[8] + (float)(i & 1); aff[1] = p[9]; asm volatile("" : "+v"(aff)); }                                    // no LDS at all
        else if (VARIANT == 9) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "=v"(aff) : "v"((unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float2 *)&s_aff[wave * 64 + ((lane + i) & 63)]) : "memory");
        else asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(aff) : "v"((unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float2 *)&s_aff[wave * 64 + ((lane + i) & 63)]) : "memory");
        f32x2 yy = {aff[1], aff[1]};
        float cz = czw[0], cw = czw[1], ay = aff[1];
#define HEAD "v_pk_mul_f32 %0, %5, %4 op_sel:[0,1]\n\tv_pk_add_f32 %3, %3, %3\n\tv_pk_fma_f32 %1, %7, %4, %0 op_sel_hi:[1,0,1]\n\t"
#define TAIL "v_pk_add_f32 %3, %3, %3\n\tv_pk_fma_f32 %2, %8, %4, %0 op_sel_hi:[1,0,1]"
#define OUTS : "=&v"(t), "=&v"(oxy), "=&v"(ozw), "+v"(filler)
#define INS  : "v"(aff), "v"(cxy), "v"(czw), "v"(vxy), "v"(vzw), "v"(yy)
        if (VARIANT >= 10) {      // no asm at all: hipcc's own packed arithmetic (vector types) against its scalar arithmetic (pinned)
            f32x2 ax = {aff[0], aff[0]}, ay2 = {aff[1], aff[1]};
            if (VARIANT == 11) asm volatile("" : "+v"(ax), "+v"(ay2));      // broadcast pairs in their own registers: no op_sel / op_sel_hi at all
            if (VARIANT == 12) asm volatile("" : "+v"(ay2));                // only the multiply's operand broadcast: op_sel_hi forms remain
            t = cxy * ay2; oxy = __builtin_elementwise_fma(vxy, ax, t);
            t2 = czw * ay2; ozw = __builtin_elementwise_fma(vzw, ax, t2);
        }
        else if (VARIANT == 0 || VARIANT == 1 || (VARIANT >= 7 && VARIANT <= 9)) asm volatile(HEAD "v_pk_mul_f32 %0, %6, %4 op_sel:[0,1]\n\t" TAIL OUTS INS);
        else if (VARIANT == 2) asm volatile(HEAD "s_nop 0\n\tv_pk_mul_f32 %0, %6, %4 op_sel:[0,1]\n\t" TAIL OUTS INS);
        else if (VARIANT == 3) asm volatile("v_pk_mul_f32 %0, %6, %5 op_sel:[0,1]\n\tv_pk_add_f32 %3, %3, %3\n\tv_pk_fma_f32 %1, %8, %5, %0 op_sel_hi:[1,0,1]\n\t"
                                            "v_pk_mul_f32 %4, %7, %5 op_sel:[0,1]\n\tv_pk_add_f32 %3, %3, %3\n\tv_pk_fma_f32 %2, %9, %5, %4 op_sel_hi:[1,0,1]"
                                            : "=&v"(t), "=&v"(oxy), "=&v"(ozw), "+v"(filler), "=&v"(t2) : "v"(aff), "v"(cxy), "v"(czw), "v"(vxy), "v"(vzw));
        else if (VARIANT == 4) {
            float tl, th;
            asm volatile("v_pk_mul_f32 %0, %5, %4 op_sel:[0,1]\n\tv_pk_add_f32 %3, %3, %3\n\tv_pk_fma_f32 %1, %7, %4, %0 op_sel_hi:[1,0,1]"
                         OUTS INS);
            // (two scalar multiplies into a fresh pair, then the second fma: what the fixed kernel does)
            tl = cz * ay; th = cw * ay;
            asm volatile("" : "+v"(tl), "+v"(th));
            f32x2 tt = {tl, th};
            asm volatile("v_pk_add_f32 %1, %1, %1\n\tv_pk_fma_f32 %0, %2, %3, %4 op_sel_hi:[1,0,1]" : "=&v"(ozw), "+v"(filler) : "v"(vzw), "v"(aff), "v"(tt));
        }
        else if (VARIANT == 5) asm volatile(HEAD "v_pk_add_f32 %3, %3, %3\n\tv_pk_mul_f32 %0, %6, %4 op_sel:[0,1]\n\t" TAIL OUTS INS);
        else asm volatile("v_pk_mul_f32 %0, %5, %9\n\tv_pk_add_f32 %3, %3, %3\n\tv_pk_fma_f32 %1, %7, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_mul_f32 %0, %6, %9\n\t" TAIL OUTS INS);
#undef HEAD
#undef TAIL
#undef OUTS
#undef INS
        float2 a2 = s_aff[wave * 64 + ((lane + i) & 63)];
        if (VARIANT == 8) a2 = make_float2(p[8] + (float)(i & 1), p[9]);
        float ex, ey, ez, ew;
        if (VARIANT >= 10) {      // scalar on purpose: every product and fma pinned into a single register
            float t0 = cxy[0] * a2.y, t1 = cxy[1] * a2.y, t2_ = czw[0] * a2.y, t3 = czw[1] * a2.y;
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2_), "+v"(t3));
            ex = __builtin_fmaf(vxy[0], a2.x, t0); asm volatile("" : "+v"(ex));
            ey = __builtin_fmaf(vxy[1], a2.x, t1); asm volatile("" : "+v"(ey));
            ez = __builtin_fmaf(vzw[0], a2.x, t2_); asm volatile("" : "+v"(ez));
            ew = __builtin_fmaf(vzw[1], a2.x, t3); asm volatile("" : "+v"(ew));
        } else {
            ex = __builtin_fmaf(vxy[0], a2.x, cxy[0] * a2.y); ey = __builtin_fmaf(vxy[1], a2.x, cxy[1] * a2.y);
            ez = __builtin_fmaf(vzw[0], a2.x, czw[0] * a2.y); ew = __builtin_fmaf(vzw[1], a2.x, czw[1] * a2.y);
        }
        nb[0] += oxy[0] != ex; nb[1] += oxy[1] != ey; nb[2] += ozw[0] != ez; nb[3] += ozw[1] != ew;
        if (filler[0] == 12345.678f) nb[0] += 1000000;
    }
    for (int c = 0; c < 4; ++c) if (nb[c]) atomicAdd(&bad[4 * (lane >> 4) + c], nb[c]);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, grid = 512;
    float *in; unsigned *bad;
    const size_t n = (size_t)grid * 512 * 12;
    CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&bad, 128));
    float *h = (float *)malloc(n * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
    CHECK(hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice));
    const char *names[] = {"0 fma reads T, next packed multiply writes T", "1 the same, no MFMA waves next to it", "2 s_nop 0 in between",
                           "3 second multiply into another pair", "4 second product by two scalar multiplies", "5 an independent packed add in between",
                           "6 as 0 without op_sel on the multiplies", "7 as 0, the pair read by compiler-generated LDS code", "8 as 0, the pair from registers (no LDS)",
                           "9 as 0, s_nop 7 x 2 behind the LDS wait", "10 NO asm: hipcc's packed code vs its scalar code", "11 NO asm, broadcast pairs made first (no op_sel, no op_sel_hi)",
                           "12 NO asm, multiplier pair made first (op_sel_hi only)"};
    for (int v = 0; v < 13; ++v) {
        CHECK(hipMemset(bad, 0, 128));
        switch (v) {
        case 0: hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 1: hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 2: hipLaunchKernelGGL(k_probe<2>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 3: hipLaunchKernelGGL(k_probe<3>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 4: hipLaunchKernelGGL(k_probe<4>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 5: hipLaunchKernelGGL(k_probe<5>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 6: hipLaunchKernelGGL(k_probe<6>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 7: hipLaunchKernelGGL(k_probe<7>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 8: hipLaunchKernelGGL(k_probe<8>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 9: hipLaunchKernelGGL(k_probe<9>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 10: hipLaunchKernelGGL(k_probe<10>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        case 11: hipLaunchKernelGGL(k_probe<11>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        default: hipLaunchKernelGGL(k_probe<12>, dim3(grid), dim3(512), 0, 0, in, bad, iters); break;
        }
        CHECK(hipDeviceSynchronize());
        unsigned hb[32];
        CHECK(hipMemcpy(hb, bad, 128, hipMemcpyDeviceToHost));
        printf("%-48s mismatches (x y z w) lanes 0-15: %u %u %u %u | 16-31: %u %u %u %u | 32-47: %u %u %u %u | 48-63: %u %u %u %u   (%.3g values per cell)\n", names[v],
               hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7], hb[8], hb[9], hb[10], hb[11], hb[12], hb[13], hb[14], hb[15], (double)grid * 4 * 16 * iters);
    }
    return 0;
}
