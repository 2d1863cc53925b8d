// scan_loop.hip -- the main loop of k_scan_f32 rebuilt ingredient by ingredient, to see which one costs the matrix pipe
// what (2 workgroups of 4 waves per CU, 64 MFMAs per wave and stage, exactly the kernel's shapes):
//   bit 0  fragment reads from LDS (5 x ds_read_b128 per 16 MFMAs), issued one k-group ahead
//   bit 1  ... issued directly in front of the MFMAs that consume them (what hipcc schedules)
//   bit 2  8 x ds_write_b128 per stage
//   bit 3  workgroup barrier per stage
//   bit 4  8 x global_load_dwordx4 per stage feeding the writes (database tile streamed, query panel re-read)
//   bit 5  8 x global_load_lds_dwordx4 per stage instead (LDS-DMA: no data registers, no LDS writes)
//   bit 7  the global loads of bit 4 are issued (asm, SGPR base + 32-bit lane offset) but nothing waits for or consumes them
//   bit 8  like bit 4, but the loads are issued by hand (asm, SGPR base + 32-bit lane offset) and ONE counted s_waitcnt
//          vmcnt(8) stands in front of the stage's LDS writes
//   bit 6  ... and never wait for them (issue cost only; the data race is irrelevant here)
//   ACCA   accumulators in the AccVGPR half of the register file (inline-asm MFMAs with "a" constraints)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/scan_loop.hip -o .variants/scan_loop && .variants/scan_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int DPAD = 512, KT = DPAD / 32, NROWS = 40064;

__device__ __forceinline__ int swz(int r, int c) { return r * 32 + 4 * (c ^ ((r >> 1) & 7)); }

template <int MODE, bool ACCA>
__global__ __launch_bounds__(256, 2) void k_loop(const float *__restrict__ g, float *out, int tiles, unsigned spread) {
    __shared__ __attribute__((aligned(16))) float s_tile[2][2][4096];
    __shared__ float s_pad[2048];                       // same LDS footprint as the kernel (72 KB): 2 workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    constexpr bool RD = MODE & 3, PRE = MODE & 1, WR = (MODE & 4) != 0, BAR = (MODE & 8) != 0, GL = (MODE & 16) != 0, DMA = (MODE & 32) != 0;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)&s_tile[0][0][0];
    const unsigned wa = lds0 + 4u * (unsigned)swz(tid >> 3, tid & 7);
    unsigned aq[4], ax[4];
    for (int u = 0; u < 4; ++u) {
        aq[u] = lds0 + 4u * (unsigned)swz(32 * wave + l31, 2 * u + h);
        ax[u] = lds0 + 4u * (unsigned)swz(l31, 2 * u + h);
    }
    for (int i = tid; i < 2 * 2 * 4096; i += 256) (&s_tile[0][0][0])[i] = 1.0f + i;
    if (tid == 0) s_pad[0] = 0.0f;
    __syncthreads();
    f32x16 acc0, acc1, acc2, acc3;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0; acc1[e] = 0; acc2[e] = 0; acc3[e] = 0; }
    const f32x4 one = {1.0f, 2.0f, 3.0f, 4.0f};
    f32x4 fa_b = one, fa_0 = one, fa_1 = one, fa_2 = one, fa_3 = one, fb_b = one, fb_0 = one, fb_1 = one, fb_2 = one, fb_3 = one;
    f32x4 ra_0 = one, ra_1 = one, ra_2 = one, ra_3 = one, ra_4 = one, ra_5 = one, ra_6 = one, ra_7 = one;
    f32x4 rb_0 = one, rb_1 = one, rb_2 = one, rb_3 = one, rb_4 = one, rb_5 = one, rb_6 = one, rb_7 = one;
    f32x4 d0, d1, d2, d3, d4, d5, d6, d7;               // dummy destinations (bit 7)
    const unsigned voff = (unsigned)(((tid >> 3) * DPAD + 4 * (tid & 7)) * 4);
    const float *qbase = g + (size_t)(blockIdx.x % spread) * 128 * DPAD;
    const int t0 = (int)((blockIdx.x * 61u) % spread);
    int lk = 0, lj = 0;
#define LD(src_, i_) (*reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>((src_) + (size_t)32 * (i_) * DPAD) + voff))
#define LDA(V, src_, i_) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(V) : "v"(voff), "s"((src_) + (size_t)32 * (i_) * DPAD) : "memory")
#define ISSUE(S)                                                                                       \
    do {                                                                                               \
        if (MODE & 128) {                                                                              \
            const float *qs_ = qbase + lk * 32;                                                        \
            const float *xs_ = g + (size_t)((t0 + lj) % spread) * 128 * DPAD + lk * 32;                   \
            LDA(d0, qs_, 0); LDA(d1, qs_, 1); LDA(d2, qs_, 2); LDA(d3, qs_, 3);                        \
            LDA(d4, xs_, 0); LDA(d5, xs_, 1); LDA(d6, xs_, 2); LDA(d7, xs_, 3);                        \
            if (++lk == KT) { lk = 0; ++lj; }                                                          \
        }                                                                                              \
        if (MODE & 256) {                                                                              \
            const float *qs_ = qbase + lk * 32;                                                        \
            const float *xs_ = g + (size_t)((t0 + lj) % spread) * 128 * DPAD + lk * 32;                \
            LDA(r##S##_0, qs_, 0); LDA(r##S##_1, qs_, 1); LDA(r##S##_2, qs_, 2); LDA(r##S##_3, qs_, 3); \
            LDA(r##S##_4, xs_, 0); LDA(r##S##_5, xs_, 1); LDA(r##S##_6, xs_, 2); LDA(r##S##_7, xs_, 3); \
            if (++lk == KT) { lk = 0; ++lj; }                                                          \
        }                                                                                              \
        if (GL) {                                                                                      \
            const float *qs_ = qbase + lk * 32;                                                        \
            const float *xs_ = g + (size_t)((t0 + lj) % spread) * 128 * DPAD + lk * 32;                   \
            r##S##_0 = LD(qs_, 0); r##S##_1 = LD(qs_, 1); r##S##_2 = LD(qs_, 2); r##S##_3 = LD(qs_, 3); \
            r##S##_4 = LD(xs_, 0); r##S##_5 = LD(xs_, 1); r##S##_6 = LD(xs_, 2); r##S##_7 = LD(xs_, 3); \
            if (++lk == KT) { lk = 0; ++lj; }                                                          \
        }                                                                                              \
    } while (0)
#define DMA1(src_, i_, BUF, OP)                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((src_) + (size_t)(32 * wave + 8 * (i_) + (lane >> 3)) * DPAD + \
                                         4 * ((lane & 7) ^ (((8 * (i_) + (lane >> 3)) >> 1) & 7))),        \
                                     (__attribute__((address_space(3))) void *)(&s_tile[BUF][OP][(32 * wave + 8 * (i_)) * 32]), 16, 0, 0)
#define DMA_STAGE(BUF)                                                                                 \
    do {                                                                                               \
        if (DMA) {                                                                                     \
            const float *qs_ = qbase + lk * 32;                                                        \
            const float *xs_ = g + (size_t)((t0 + lj) % spread) * 128 * DPAD + lk * 32;                   \
            DMA1(qs_, 0, BUF, 0); DMA1(qs_, 1, BUF, 0); DMA1(qs_, 2, BUF, 0); DMA1(qs_, 3, BUF, 0);    \
            DMA1(xs_, 0, BUF, 1); DMA1(xs_, 1, BUF, 1); DMA1(xs_, 2, BUF, 1); DMA1(xs_, 3, BUF, 1);    \
            if (++lk == KT) { lk = 0; ++lj; }                                                          \
        }                                                                                              \
    } while (0)
#define ST1(V, OFF) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(wa), "v"(V), "n"(OFF) : "memory")
#define COMMIT(S, BUF)                                                                                 \
    do {                                                                                               \
        if (WR) {                                                                                      \
            if (MODE & 256) asm volatile("s_waitcnt vmcnt(8)" : "+v"(r##S##_0), "+v"(r##S##_1), "+v"(r##S##_2), "+v"(r##S##_3), \
                                         "+v"(r##S##_4), "+v"(r##S##_5), "+v"(r##S##_6), "+v"(r##S##_7));  \
            ST1(r##S##_0, (BUF) * 32768); ST1(r##S##_1, (BUF) * 32768 + 4096); ST1(r##S##_2, (BUF) * 32768 + 8192); \
            ST1(r##S##_3, (BUF) * 32768 + 12288); ST1(r##S##_4, (BUF) * 32768 + 16384); ST1(r##S##_5, (BUF) * 32768 + 20480); \
            ST1(r##S##_6, (BUF) * 32768 + 24576); ST1(r##S##_7, (BUF) * 32768 + 28672);                \
        }                                                                                              \
    } while (0)
#define FRAG(F, BUF, U)                                                                                \
    do { if (RD)                                                                                       \
    asm volatile("ds_read_b128 %0, %5 offset:%7\n\tds_read_b128 %1, %6 offset:%8\n\tds_read_b128 %2, %6 offset:%9\n\t" \
                 "ds_read_b128 %3, %6 offset:%10\n\tds_read_b128 %4, %6 offset:%11"                    \
                 : "=&v"(F##_b), "=&v"(F##_0), "=&v"(F##_1), "=&v"(F##_2), "=&v"(F##_3)                \
                 : "v"(aq[U]), "v"(ax[U]), "n"((BUF) * 32768), "n"((BUF) * 32768 + 16384),             \
                   "n"((BUF) * 32768 + 20480), "n"((BUF) * 32768 + 24576), "n"((BUF) * 32768 + 28672) : "memory"); } while (0)
#define WAIT(F, N) do { if (RD) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F##_b), "+v"(F##_0), "+v"(F##_1), "+v"(F##_2), "+v"(F##_3)); } while (0)
#define MFMA_A(ACC, A_, B_) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A_), "v"(B_))
#define MFMA(F)                                                                                        \
    do {                                                                                               \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                \
            if (ACCA) {                                                                                \
                MFMA_A(acc0, F##_0[m], F##_b[m]); MFMA_A(acc1, F##_1[m], F##_b[m]);                    \
                MFMA_A(acc2, F##_2[m], F##_b[m]); MFMA_A(acc3, F##_3[m], F##_b[m]);                    \
                continue;                                                                              \
            }                                                                                          \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_0[m], F##_b[m], acc0, 0, 0, 0);            \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_1[m], F##_b[m], acc1, 0, 0, 0);            \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_2[m], F##_b[m], acc2, 0, 0, 0);            \
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_3[m], F##_b[m], acc3, 0, 0, 0);            \
        }                                                                                              \
    } while (0)
#define BARRIER() do { if (DMA && !(MODE & 64)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
                       if (ACCA) { if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3) : : "memory"); \
                                   else asm volatile("" : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3)); }  \
                       else if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : : "memory"); \
                       else asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3)); } while (0)
    // prefetched form (variant B of the kernel) / exposed form (reads directly in front of their MFMAs, barrier at the end)
#define STEP(BUF, SET)                                                                                 \
    do {                                                                                               \
        if (PRE) {                                                                                     \
            FRAG(fb, BUF, 1); WAIT(fa, 5); MFMA(fa);                                                   \
            FRAG(fa, BUF, 2); WAIT(fb, 5); MFMA(fb);                                                   \
            COMMIT(SET, (BUF) ^ 1); ISSUE(SET);                                                        \
            FRAG(fb, BUF, 3); if (WR) WAIT(fa, 13); else WAIT(fa, 5);                                  \
            MFMA(fa);                                                                                  \
            BARRIER();                                                                                 \
            DMA_STAGE(BUF);          /* this buffer's reads are complete: refill it, one full stage to land */ \
            FRAG(fa, (BUF) ^ 1, 0); WAIT(fb, 5); MFMA(fb);                                             \
        } else {                                                                                       \
            FRAG(fa, BUF, 0); WAIT(fa, 0); MFMA(fa);                                                   \
            FRAG(fb, BUF, 1); WAIT(fb, 0); MFMA(fb);                                                   \
            COMMIT(SET, (BUF) ^ 1); ISSUE(SET); DMA_STAGE((BUF) ^ 1);                                  \
            FRAG(fa, BUF, 2); WAIT(fa, 0); MFMA(fa);                                                   \
            FRAG(fb, BUF, 3); WAIT(fb, 0); MFMA(fb);                                                   \
            BARRIER();                                                                                 \
        }                                                                                              \
    } while (0)
    ISSUE(a); ISSUE(b);
    if (PRE) FRAG(fa, 0, 0);
    for (int it = 0; it < tiles * KT; it += 2) {
        STEP(0, a);
        STEP(1, b);
    }
    WAIT(fa, 0); WAIT(fb, 0);
    if (MODE & 256) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra_0), "+v"(ra_7), "+v"(rb_0), "+v"(rb_7));
    if (MODE & 128) asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
    float s = 0.0f;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    s += ra_0[0] + rb_0[0] + ra_7[3] + rb_7[3];
    if (s == 123.456f) out[0] = s + s_pad[0];
}

// What does a co-resident wave's VALU work (a filter epilogue) cost the other wave's MFMA loop?  Workgroups in the odd
// wave slot run `valu_per_mfma` dependent-free v_add_f32 per ... nothing else; those in the even slot run the loop.
__global__ __launch_bounds__(256, 2) void k_valu_partner(float *out, unsigned long long *lives, int iters, int mfma_only) {
    const unsigned slot = __builtin_amdgcn_s_getreg((1 << 11) | 4) & 1u;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    float x0 = threadIdx.x, x1 = 1.0f, x2 = 2.0f, x3 = 3.0f;
    f32x16 acc0, acc1, acc2, acc3;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0; acc1[e] = 0; acc2[e] = 0; acc3[e] = 0; }
    if (slot == 0 || mfma_only) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc3, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it)       // 64 MFMAs = 4096 cycles on the other wave; 1024 VALU = 4096 issue cycles here
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t.endr"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(1.0f));
    }
    float s = x0 + x1 + x2 + x3;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) {
        atomicAdd(&lives[2 * slot], __builtin_amdgcn_s_memrealtime() - rt0);
        atomicAdd(&lives[2 * slot + 1], 1ull);
    }
}

// ---- the same loop with 16-wide stages (64-B LDS rows, 32 KB of tiles per workgroup): three workgroups per CU ----
__device__ __forceinline__ int swz16(int r, int c) { return r * 16 + 4 * (c ^ ((r >> 2) & 3)); }
template <int WGS>
__global__ __launch_bounds__(256, WGS) void k_loop16(const float *__restrict__ g, float *out, int tiles, unsigned spread) {
    __shared__ __attribute__((aligned(16))) float s_tile[2][2][2048];    // [buf][Q|X][128 rows x 16]
    __shared__ float s_pad[WGS == 3 ? 2048 : 8192];                       // 40 KB (3 per CU) / 64 KB (2 per CU)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)&s_tile[0][0][0];
    const unsigned wa = lds0 + 4u * (unsigned)swz16(tid >> 2, tid & 3);   // row tid>>2 (+64), chunk tid&3
    unsigned aq[2], ax[2];
    for (int u = 0; u < 2; ++u) {
        aq[u] = lds0 + 4u * (unsigned)swz16(32 * wave + l31, 2 * u + h);
        ax[u] = lds0 + 4u * (unsigned)swz16(l31, 2 * u + h);
    }
    for (int i = tid; i < 2 * 2 * 2048; i += 256) (&s_tile[0][0][0])[i] = 1.0f + i;
    if (tid == 0) s_pad[0] = 0.0f;
    __syncthreads();
    f32x16 acc0, acc1, acc2, acc3;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0; acc1[e] = 0; acc2[e] = 0; acc3[e] = 0; }
    const f32x4 one = {1.0f, 2.0f, 3.0f, 4.0f};
    f32x4 fa_b = one, fa_0 = one, fa_1 = one, fa_2 = one, fa_3 = one, fb_b = one, fb_0 = one, fb_1 = one, fb_2 = one, fb_3 = one;
    f32x4 ra_0 = one, ra_1 = one, ra_2 = one, ra_3 = one, rb_0 = one, rb_1 = one, rb_2 = one, rb_3 = one;
    const unsigned voff = (unsigned)(((tid >> 2) * DPAD + 4 * (tid & 3)) * 4), voffb = voff + 64u * DPAD * 4;
    const float *qbase = g + (size_t)(blockIdx.x % spread) * 128 * DPAD;
    const int t0 = (int)((blockIdx.x * 61u) % spread);
    int lk = 0, lj = 0;
    constexpr int KT16 = DPAD / 16;
#define ISSUE16(S)                                                                                     \
    do {                                                                                               \
        const float *qs_ = qbase + lk * 16;                                                            \
        const float *xs_ = g + (size_t)((t0 + lj) % spread) * 128 * DPAD + lk * 16;                    \
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %5, %6\n\t"  \
                     "global_load_dwordx4 %2, %4, %7\n\tglobal_load_dwordx4 %3, %5, %7"                \
                     : "=&v"(r##S##_0), "=&v"(r##S##_1), "=&v"(r##S##_2), "=&v"(r##S##_3)              \
                     : "v"(voff), "v"(voffb), "s"(qs_), "s"(xs_) : "memory");                          \
        if (++lk == KT16) { lk = 0; ++lj; }                                                            \
    } while (0)
#define ST16(V, OFF) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(wa), "v"(V), "n"(OFF) : "memory")
#define COMMIT16(S, BUF)                                                                               \
    do {                                                                                               \
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(r##S##_0), "+v"(r##S##_1), "+v"(r##S##_2), "+v"(r##S##_3)); \
        ST16(r##S##_0, (BUF) * 16384); ST16(r##S##_1, (BUF) * 16384 + 4096);                           \
        ST16(r##S##_2, (BUF) * 16384 + 8192); ST16(r##S##_3, (BUF) * 16384 + 12288);                   \
    } while (0)
#define FRAG16(F, BUF, U)                                                                              \
    asm volatile("ds_read_b128 %0, %5 offset:%7\n\tds_read_b128 %1, %6 offset:%8\n\tds_read_b128 %2, %6 offset:%9\n\t" \
                 "ds_read_b128 %3, %6 offset:%10\n\tds_read_b128 %4, %6 offset:%11"                    \
                 : "=&v"(F##_b), "=&v"(F##_0), "=&v"(F##_1), "=&v"(F##_2), "=&v"(F##_3)                \
                 : "v"(aq[U]), "v"(ax[U]), "n"((BUF) * 16384), "n"((BUF) * 16384 + 8192),              \
                   "n"((BUF) * 16384 + 10240), "n"((BUF) * 16384 + 12288), "n"((BUF) * 16384 + 14336) : "memory")
#define WAIT16(F, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F##_b), "+v"(F##_0), "+v"(F##_1), "+v"(F##_2), "+v"(F##_3))
#define MFMA16(F)                                                                                      \
    do {                                                                                               \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_0[m], F##_b[m], acc0, 0, 0, 0);            \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_1[m], F##_b[m], acc1, 0, 0, 0);            \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_2[m], F##_b[m], acc2, 0, 0, 0);            \
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_3[m], F##_b[m], acc3, 0, 0, 0);            \
        }                                                                                              \
    } while (0)
    // stage: fa holds k-group 0; reads of group 1 ahead; commit + issue; barrier; next stage's group 0
#define STEP16(BUF, SET)                                                                               \
    do {                                                                                               \
        FRAG16(fb, BUF, 1); WAIT16(fa, 5); MFMA16(fa);                                                 \
        COMMIT16(SET, (BUF) ^ 1); ISSUE16(SET);                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : : "memory"); \
        FRAG16(fa, (BUF) ^ 1, 0); WAIT16(fb, 5); MFMA16(fb);                                           \
    } while (0)
    ISSUE16(a); ISSUE16(b);
    FRAG16(fa, 0, 0);
    for (int it = 0; it < tiles * KT16; it += 2) { STEP16(0, a); STEP16(1, b); }
    WAIT16(fa, 0); WAIT16(fb, 0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra_0), "+v"(ra_3), "+v"(rb_0), "+v"(rb_3));
    float s = 0.0f;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    s += ra_0[0] + rb_0[0];
    if (s == 123.456f) out[0] = s + s_pad[0];
}

template <int WGS>
static void run16(const float *g, float *out, int cus, const char *what) {
    const int tiles = 120, grid = WGS * cus;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_loop16<WGS>), dim3(grid), dim3(256), 0, 0, g, out, tiles, 300u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_loop16<WGS>), dim3(grid), dim3(256), 0, 0, g, out, tiles, 300u);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double flop = (double)grid * 4 * tiles * (DPAD / 16) * 32 * 4096;
    printf("16-wide stages, %d workgroups per CU: %-50s %7.3f ms  %6.1f TFLOP/s  %.3f of 157.3\n", WGS, what, best, flop / best / 1e9,
           flop / best / 1e9 / 157.3);
}

template <int MODE, bool ACCA = false>
static void run(const float *g, float *out, int cus, const char *what, unsigned spread = 300, int wg_per_cu = 2) {
    const int tiles = 120, grid = wg_per_cu * cus;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_loop<MODE, ACCA>), dim3(grid), dim3(256), 0, 0, g, out, tiles, spread);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_loop<MODE, ACCA>), dim3(grid), dim3(256), 0, 0, g, out, tiles, spread);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double flop = (double)grid * 4 * tiles * KT * 64 * 4096;
    printf("mode %2d%s %-68s %7.3f ms  %6.1f TFLOP/s  %.3f of 157.3\n", MODE, ACCA ? "a" : " ", what, best, flop / best / 1e9, flop / best / 1e9 / 157.3);
}

int main() {
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *g, *out;
    CHECK(hipMalloc(&g, (size_t)NROWS * DPAD * 4)); CHECK(hipMemset(g, 0, (size_t)NROWS * DPAD * 4));
    CHECK(hipMalloc(&out, 64));
    run<0>(g, out, cus, "MFMAs on registers only");
    run<1>(g, out, cus, "+ fragment reads, one k-group ahead");
    run<2>(g, out, cus, "+ fragment reads, exposed");
    run<1 | 8>(g, out, cus, "+ reads ahead + barrier");
    run<2 | 8>(g, out, cus, "+ reads exposed + barrier");
    run<1 | 4 | 8>(g, out, cus, "+ reads ahead + LDS writes + barrier");
    run<2 | 4 | 8>(g, out, cus, "+ reads exposed + LDS writes + barrier");
    run<1 | 4 | 8 | 16>(g, out, cus, "+ reads ahead + global loads + LDS writes + barrier (= the kernel's loop)");
    run<2 | 4 | 8 | 16>(g, out, cus, "+ reads exposed + global loads + LDS writes + barrier");
    run<4 | 8 | 16>(g, out, cus, "no fragment reads: global loads + LDS writes + barrier");
    run<1 | 8 | 32>(g, out, cus, "+ reads ahead + LDS-DMA + barrier");
    run<1 | 4 | 8 | 128>(g, out, cus, "+ reads ahead + LDS writes + barrier + global loads nobody waits for");
    run<1 | 4 | 8 | 16>(g, out, cus, "the kernel's loop, every workgroup on the same 4 tiles (L2 hits)", 4);
    run<1 | 4 | 8 | 16>(g, out, cus, "the kernel's loop, 32 tiles (8 MB footprint)", 32);
    run<1 | 4 | 8 | 256>(g, out, cus, "the kernel's loop with hand-issued global loads and one counted vmcnt(8) per stage");
    run<1 | 4 | 8 | 256>(g, out, cus, "the same with ONE workgroup per CU (what a wave achieves alone)", 300, 1);
    run<2 | 4 | 8 | 16>(g, out, cus, "hipcc-style loop with ONE workgroup per CU", 300, 1);
    run<0>(g, out, cus, "MFMAs on registers only, ONE workgroup per CU", 300, 1);
    run<1 | 8 | 32 | 64>(g, out, cus, "+ reads ahead + LDS-DMA never waited for + barrier");
    run<8 | 32 | 64>(g, out, cus, "no fragment reads: LDS-DMA never waited for + barrier");
    run<1 | 4 | 8 | 16, true>(g, out, cus, "AccVGPR accumulators: the kernel's loop");
    run16<2>(g, out, cus, "hand-issued loop");
    run16<3>(g, out, cus, "hand-issued loop");
    {
        unsigned long long *lives, h[4];
        CHECK(hipMalloc(&lives, 32));
        const int iters = 4000;
        for (int mfma_only = 1; mfma_only >= 0; --mfma_only) {
            CHECK(hipMemset(lives, 0, 32));
            hipLaunchKernelGGL(k_valu_partner, dim3(2 * cus), dim3(256), 0, 0, out, lives, iters, mfma_only);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, lives, 32, hipMemcpyDeviceToHost));
            printf("%s: even-slot workgroups (MFMA loop, %d x 64 MFMAs) live %.3f ms on average (%llu), odd-slot %.3f ms (%llu)\n",
                   mfma_only ? "both workgroups of a CU run MFMAs        " : "odd-slot workgroups run v_add_f32 instead",
                   iters, h[1] ? h[0] / 1e5 / h[1] : 0.0, h[1], h[3] ? h[2] / 1e5 / h[3] : 0.0, h[3]);
        }
    }
    return 0;
}
