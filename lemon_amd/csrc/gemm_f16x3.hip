// gemm_f16x3.hip -- hand-written split-fp16 GEMM for the MLP of a transformer block (gfx950 only).
//
//   y = act(alpha * x W^T + bias) [+ residual]        fp32 x [m, k], W [n, k] carried as fp16 pairs, fp32 accumulate
//
// the arithmetic of lemon_linear_f16x3 (linear.hip: hi.hi + hi.lo + lo.hi on the fp16 matrix cores) in a kernel that knows the
// operand structure, which the library kernel behind lemon_linear_f16x3 cannot:
//   * operands are pre-packed TILE-MAJOR in MFMA fragment order (split3.hpp: tiled_off): a stage of the LDS ring is one
//     contiguous global_load_lds copy, fragment reads are linear ds_read_b128 (no swizzle, no bank conflicts);
//   * the activation's hi part is staged once (the library's [hi | hi | lo] row stages it twice) and W_hi 2^-11 is made in
//     registers (four exact v_pk_mul_f16 per fragment): 24 KB per k16 step and workgroup instead of 36 KB;
//   * the epilogue is ours: fc1's bias + QuickGELU (SiLU with the scale folded into alpha / bias) + the fp16 split are applied
//     to the accumulators and stored AS THE NEXT GEMM'S OPERAND -- the MFMA output layout and the tile-major operand layout
//     coincide, a wave's store instruction covers 512 contiguous bytes -- so the [m, mlp] fp32 activation tensor and the split
//     pass over it (k_split3_rows: 7 % of the headline step) do not exist any more.
// Workgroup: 128 (m) x 256 (n) output tile, four waves of 64 x 128 (8 accumulator tiles = 128 AccVGPRs), two workgroups per
// CU (one's epilogue runs under the other's main loop), one k16 step per stage (A_hi, A_lo: 2 x 4 KB, W_hi, W_lo: 2 x 8 KB),
// ring of 3 stages, 24 MFMAs per wave per barrier.  Tiles are walked in super-blocks of 32 m-tiles x 1 n-tile per XCD: the
// weight tile column (393 KB per 768 k) stays in that XCD's L2, the activation rows stream.
// Measured against the library on random operands (tools/micro/gemm_f16x3.hip, profiles/r3/micro_gemm_f16x3.txt): plain
// epilogue 0.95-0.98 of the library's rate at the MLP shapes; fc1 with the fused epilogue 772 us against 669 + 264 us for
// library GEMM + split pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "split3.hpp"

using namespace lemon_split;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

constexpr int TM = TILE_A_ROWS, TN = TILE_W_ROWS;       // 128 x 256
constexpr int IB = TM / 64;                             // 32-row activation blocks per wave (2)
constexpr int BLKA = TM * 32, BLKW = TN * 32;           // bytes of one segment of one k16 step: 4 KB, 8 KB
constexpr int STAGE = 2 * BLKA + 2 * BLKW;              // 24 KB
constexpr int NB = 3, LA = NB - 1;
#ifndef LEMON_GEMM_GM
#define LEMON_GEMM_GM 32
#endif
constexpr int GM = LEMON_GEMM_GM;                       // m-tiles per super-block (32 = half the workgroups an XCD runs at a time; 16 / 64 measured no better)
constexpr int DMA_PER_STAGE = BLKA / 2048 + 4;          // 1-KB instructions per wave and stage (2 + 4)

struct GemmParams {
    const char *at, *wt;       // tile-major operands
    const float *bias;         // [n] or null
    const float *residual;     // [m, n] fp32 or null (EPI 0)
    void *out;                 // EPI 0: fp32 [m, n]; EPI 1 / 2: tile-major activation operand of the next GEMM (its k = n)
    int64_t m;
    int n, ks, m_tiles, n_tiles;
    int gm, gn;                // super-block of the tile walk: gm m-tiles x gn n-tiles per XCD at a time
    int vblocks;               // virtual workgroup ids of the tile walk (>= tiles: ragged super-blocks leave holes)
    float alpha;
    // LayerNorm folded into the GEMM (k_gemm_f16x3t16 only; see "LayerNorm fold" above lemon_linear_f16x3t_ln):
    const float2 *row_aff;     // FOLD: per row (rstd, -mean rstd) of the LayerNorm in front of this GEMM, or null
    const float *colsum;       // FOLD: [n] alpha * sum_k of the packed weight row (the weight carries the LayerNorm gain)
    const unsigned short *residual_t;   // EMIT: the residual as a tile-major activation operand [m, n] (hi + lo 2^-11) instead of fp32, or null
    unsigned short *emit_t;    // EMIT (EPI 0): the fp32 result also as the tile-major operand of the next GEMM (its k = n), or null
    float *emit_stats;         // EMIT: [m][2 n_tiles][2] per row and 128-column group (mean, sum of squared deviations)
#ifdef LEMON_GEMM_PHASES
    unsigned long long *dbg;   // diagnostic build: summed shader cycles of [start -> first barrier passed, main loop, epilogue], workgroups,
                               // 100-MHz ticks resident, in-loop cycles of [DMA wait + B, reads + block 0 + B', blocks 1-7]
#endif
};

// Tile walk.  Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2); an XCD works through
// super-blocks of gm x gn tiles, m fastest: the workgroups it runs at a time (2 per CU = 64) then share gm activation tile
// rows and gn weight tile columns instead of streaming one of the two operands once per tile.
__device__ __forceinline__ bool tile_of_workgroup(const GemmParams &p, unsigned vb, int &mt, int &nt) {
    const int xcd = vb & 7, q = vb >> 3;
    const int per = p.gm * p.gn;
    const int gm_n = (p.m_tiles + p.gm - 1) / p.gm, gn_n = (p.n_tiles + p.gn - 1) / p.gn;
    const int blk = (q / per) * 8 + xcd, pos = q % per;
    if (blk >= gm_n * gn_n) return false;
    mt = (blk / gn_n) * p.gm + pos % p.gm;
    nt = (blk % gn_n) * p.gn + pos / p.gm;
    return mt < p.m_tiles && nt < p.n_tiles;
}

__device__ __forceinline__ void mfma(f32x16 &acc, const h16x8 &a, const h16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// 1 KB of LDS-DMA: `SGPR base + 32-bit lane offset`, M0 = LDS destination (nothing else in this kernel uses M0)
__device__ __forceinline__ void dma1k(const char *src, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src), "s"(lds) : "memory");
}
__device__ __forceinline__ h16x8 lds128(unsigned addr, int off) {
    h16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}

// The operand epilogues' activation.  EPI 1: SiLU u sigmoid(u) (v_exp_f32 / v_rcp_f32: 1 ulp each) -- QuickGELU with the 1.702
// folded into alpha and the bias by the caller; EPI 2: the exact GELU u Phi(u) = u/2 (1 + erf(u / sqrt 2)) of the BERT / timm
// towers (open_clip BiomedCLIP, lib/models/utils.py:72-78; ocml's erff: a few ulp)
template <int EPI>
__device__ __forceinline__ float epilogue_act(float u) {
    if (EPI == 2) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
    return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
}

// EPI 0: fp32 row-major (+ residual); EPI 1 / 2: SiLU / GELU, fp16 split, tile-major operand
template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_f16x3t(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int mt, nt;
    if (!tile_of_workgroup(p, blockIdx.x, mt, nt)) return;
    const int KS = p.ks;
    const char *a_src = p.at + (size_t)mt * KS * 2 * BLKA;
    const char *w_src = p.wt + (size_t)nt * KS * 2 * BLKW;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)smem;
    // DMA shares of a stage: wave w moves A bytes [2 KB w, +2 KB) and W bytes [4 KB w, +4 KB)
    const unsigned va = (unsigned)(wave * (BLKA / 2) + lane * 16), vw = (unsigned)(wave * (BLKW / 2) + lane * 16);
    auto issue = [&](int ks, int slot) {
        const char *as = a_src + (size_t)ks * 2 * BLKA, *ws = w_src + (size_t)ks * 2 * BLKW;
        const unsigned la = lds0 + slot * STAGE + wave * (BLKA / 2), lw = lds0 + slot * STAGE + 2 * BLKA + wave * (BLKW / 2);
#pragma unroll
        for (int j = 0; j < BLKA / 2048; ++j) dma1k(as, va + j * 1024, la + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma1k(ws, vw + j * 1024, lw + j * 1024);
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
    // fragment addresses inside a stage: A block (IB wm + i), W block (4 wn + j) of each segment
    const unsigned fa = lds0 + wm * (IB * 1024) + lane * 16, fw = lds0 + 2 * BLKA + wn * 4096 + lane * 16;
    h16x8 af[2][IB][2], wf[2][4][3];
#define READ_FRAGS(set, slot)                                                                                   \
    do {                                                                                                        \
        const unsigned pa_ = fa + (slot) * STAGE, pw_ = fw + (slot) * STAGE;                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) {                                                     \
            af[set][i_][0] = lds128(pa_, i_ * 1024); af[set][i_][1] = lds128(pa_, BLKA + i_ * 1024);            \
        }                                                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
            wf[set][j_][0] = lds128(pw_, j_ * 1024); wf[set][j_][1] = lds128(pw_, BLKW + j_ * 1024);            \
        }                                                                                                       \
    } while (0)
#define DO_MFMAS(set)                                                                                           \
    do {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) wf[set][j_][2] = wf[set][j_][0] * (_Float16)0.00048828125f;      /* W_hi 2^-11: exact */ \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) mfma(acc[j_][i_], wf[set][j_][0], af[set][i_][0]); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) mfma(acc[j_][i_], wf[set][j_][1], af[set][i_][0]); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) mfma(acc[j_][i_], wf[set][j_][2], af[set][i_][1]); \
    } while (0)
    // the wait for a fragment set names its registers as in/out operands: hipcc cannot see that the ds_read asm statements
    // deliver late, and must not move a use of these registers (the v_pk_mul_f16 above are ordinary code) in front of the wait
#define WAIT_FRAGS(set)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                         \
                 : "+v"(af[set][0][0]), "+v"(af[set][0][1]), "+v"(af[set][1][0]), "+v"(af[set][1][1]),          \
                   "+v"(wf[set][0][0]), "+v"(wf[set][0][1]), "+v"(wf[set][1][0]), "+v"(wf[set][1][1]),          \
                   "+v"(wf[set][2][0]), "+v"(wf[set][2][1]), "+v"(wf[set][3][0]), "+v"(wf[set][3][1]) : : "memory")
    static_assert(IB == 2, "WAIT_FRAGS lists two activation blocks");
    static_assert((LA - 1) * DMA_PER_STAGE == 6, "the counted wait below");
#define WAIT_STAGE() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")      /* all but the youngest stage in flight have landed */
#define PIN_ACC() do { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_) asm volatile("" : "+a"(acc[j_][i_])); } while (0)
    if (KS > 1) WAIT_STAGE(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    READ_FRAGS(0, 0);
    WAIT_FRAGS(0);
    PIN_ACC();
    for (int t = 0; t < KS; t += 2) {
        PIN_ACC();          // (the loop-carried accumulators stay AccVGPRs: left alone hipcc homes them in VGPRs and copies them every step)
        // ---- step t (fragment set 0); stage t+1's fragments (set 1) are read under its MFMAs ----
        if (t + LA < KS) issue(t + LA, (t + LA) % NB);       // the slot of stage t-1: everyone is past the barrier behind its reads
        if (t + 1 < KS) {
            if (t + LA < KS) WAIT_STAGE(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            READ_FRAGS(1, (t + 1) % NB);
        }
        DO_MFMAS(0);
        WAIT_FRAGS(1);
        if (t + 1 >= KS) break;
        // ---- step t+1 (set 1) ----
        if (t + 1 + LA < KS) issue(t + 1 + LA, (t + 1 + LA) % NB);
        if (t + 2 < KS) {
            if (t + 1 + LA < KS) WAIT_STAGE(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            READ_FRAGS(0, (t + 2) % NB);
        }
        DO_MFMAS(1);
        WAIT_FRAGS(0);
    }
#undef WAIT_FRAGS
#undef READ_FRAGS
#undef DO_MFMAS
#undef WAIT_STAGE
#undef PIN_ACC
    // ---- epilogue: the lane holds row m = l%32 of activation block i and columns n = 8 g + 4 (l/32) + e of weight block j ----
    const int l31 = lane & 31, h = lane >> 5;
    const int N = p.n;
    if (EPI == 0) {
        // fp32 row-major result (+ residual): in the accumulator layout a store instruction would touch 32 rows with 32 bytes
        // each.  Every 32 x 32 accumulator tile therefore goes through a wave-private LDS patch (the ring is free now) and
        // leaves as full 128-byte lines: 8 lanes per row, 8 rows per instruction, residual loads the same way.
        __syncthreads();                                               // every wave is done with the ring
        float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 36);
        const int prow = lane >> 3, pcol = 4 * (lane & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nt * TN + wn * 128 + j * 32 + pcol;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) b = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int64_t m0 = (int64_t)mt * TM + wm * (IB * 32) + i * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(patch + l31 * 36 + 8 * g + 4 * h) =
                        make_float4(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
                float4 r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int64_t m = m0 + 8 * q + prow;
                    if (p.residual && m < p.m) r[q] = *reinterpret_cast<const float4 *>(p.residual + m * N + n);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t m = m0 + 8 * q + prow;
                    const float4 v = *reinterpret_cast<const float4 *>(patch + (8 * q + prow) * 36 + pcol);
                    if (m < p.m)
                        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.out) + m * N + n) =
                            make_float4(p.alpha * v.x + b.x + r[q].x, p.alpha * v.y + b.y + r[q].y, p.alpha * v.z + b.z + r[q].z, p.alpha * v.w + b.w + r[q].w);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int64_t m = (int64_t)mt * TM + wm * (IB * 32) + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = nt * TN + wn * 128 + j * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + 8 * g;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) b = *reinterpret_cast<const float4 *>(p.bias + n);
                const float o[4] = {p.alpha * acc[j][i][4 * g] + b.x, p.alpha * acc[j][i][4 * g + 1] + b.y,
                                    p.alpha * acc[j][i][4 * g + 2] + b.z, p.alpha * acc[j][i][4 * g + 3] + b.w};
                // the next GEMM's activation operand (its k = this n): lanes l and l+32 fill one 16-byte slot, a wave's store
                // instruction covers 512 contiguous bytes.  Rows >= m are computed from whatever the operand's pad rows
                // hold and land in the next operand's pad rows: never read into a stored result.
                h16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = epilogue_act<EPI>(o[e]);
                    unsigned short a_, b_, c_;
                    split2h<false>(v, a_, b_, c_);
                    hi[e] = __builtin_bit_cast(_Float16, a_); lo[e] = __builtin_bit_cast(_Float16, c_);
                }
                unsigned short *base = reinterpret_cast<unsigned short *>(p.out) + tiled_off(TM, m, n, 0, N);
                *reinterpret_cast<h16x4 *>(base) = hi;
                *reinterpret_cast<h16x4 *>(base + TM * 16) = lo;
            }
        }
    }
}

// ---- the same GEMM on v_mfma_f32_16x16x32_f16 --------------------------------------------------------------------------
// Same operands, same ring, same tile walk; what changes is the matrix instruction and with it the step: one MFMA now spans
// k = 32, i.e. TWO k16 slots of the ring -- a lane's 16-byte fragment piece comes from the first slot for lanes 0-31 (k groups
// 0, 1) and from the second for lanes 32-63 (k groups 2, 3), both inside the unchanged tile-major layout (a 16-row x 32-k
// fragment = rows (b & 1) 16 .. +15 of a 32-row block, k halves of two consecutive k16 steps; every 16-lane read group still
// covers 256 contiguous bytes: no bank conflicts).  Why: the board is power-limited on these GEMMs (MFMA-busy 0.55-0.6 at
// 1.7-1.85 GHz, profiles/r3/encoder_mfma_pmc.csv) and the 16x16x32 form holds a higher clock than 32x32x16 at equal cycles per
// flop (MI355X_MICROARCH.md, DVFS give-back item 7; tools/micro/mfma_peak.hip: 2 224 vs 1 875 MHz on random operands).
// Per k32 step and wave (64 x 128 of the output = 4 x 8 accumulator tiles of 16 x 16: 128 AccVGPRs): 24 fragment reads
// (A 4 blocks x hi, lo; W 8 blocks x hi, lo; in two batches: at most 15 LDS operations may be counted at once), 96 MFMAs; fragments are single-buffered (A 32 + W 64 registers; W_hi 2^-11 is
// made per block right before its four MFMAs) -- the other workgroup of the CU covers a wave's read phase.  A ring of three k16
// slots with two consumed per step needs the slots back as soon as they are read: barrier B (both slots landed, everyone's
// DMA), the reads, barrier B' (everyone's reads done) behind the first weight block's MFMAs, then the DMAs of slots t+3, t+4
// into the two positions just read, with the rest of the step's MFMAs (84 x 16 cycles, twice that wall with the partner
// workgroup) to land.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16(f32x4 &acc, const h16x8 &a, const h16x8 &b) {
#ifdef LEMON_MFMA16_BUILTIN
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
#else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
}

// LEAN (EMIT only): which of the chain's two options the instantiation is built for -- bit 0: no fp32 result (the output projection
// inside a chain), bit 1: residual in operand form (fc2 inside a chain); 0 = the classic emit form.  (One instantiation deciding both
// at run time spilled six registers in its epilogue.)
template <int EPI, bool FOLD, bool EMIT, int LEAN = 0>
__global__ __launch_bounds__(256, 2) void k_gemm_f16x3t16(GemmParams p) {
    static_assert(!EMIT || EPI == 0, "the operand + statistics output rides on the fp32 epilogue");
    static_assert(LEAN == 0 || EMIT, "the chain options belong to the emitting form");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int mt, nt;
    // (one tile per workgroup, dispatched by the hardware as slots free up.  A persistent form -- 512 workgroups looping over
    // their tiles in the same walk order -- measured 2-5 % SLOWER at the tower shapes: tools/micro/gemm_ab.hip, round 4)
    if (!tile_of_workgroup(p, blockIdx.x, mt, nt)) return;
#ifdef LEMON_GEMM_PHASES
    const unsigned long long ph0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long ph1 = 0;
#endif
    const int KS = p.ks;                                    // k16 slots: even (the host checks k % 32 == 0)
    const char *a_src = p.at + (size_t)mt * KS * 2 * BLKA;
    const char *w_src = p.wt + (size_t)nt * KS * 2 * BLKW;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)smem;
    const unsigned va = (unsigned)(wave * (BLKA / 2) + lane * 16), vw = (unsigned)(wave * (BLKW / 2) + lane * 16);
    auto issue = [&](int ks, int slot) {
        const char *as = a_src + (size_t)ks * 2 * BLKA, *ws = w_src + (size_t)ks * 2 * BLKW;
        const unsigned la = lds0 + slot * STAGE + wave * (BLKA / 2), lw = lds0 + slot * STAGE + 2 * BLKA + wave * (BLKW / 2);
#pragma unroll
        for (int j = 0; j < BLKA / 2048; ++j) dma1k(as, va + j * 1024, la + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma1k(ws, vw + j * 1024, lw + j * 1024);
    };
    f32x4 acc[8][4];                                        // [weight block c of 16 n][activation block b of 16 m]
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][b][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < NB; ++s)
        if (s < KS) issue(s, s);
    // a lane's piece of a fragment: row r16 of the 16-row block, k group kg of the k32 step = k half (kg & 1) of slot (kg >> 1)
    const int r16 = lane & 15, kg = lane >> 4;
    const unsigned lane_off = (unsigned)((kg & 1) * 512 + r16 * 16);
    const unsigned fa = lds0 + wm * (IB * 1024) + lane_off, fw = lds0 + 2 * BLKA + wn * 4096 + lane_off;
    const bool upper = lane >= 32;
    h16x8 af[4][2], wf[8][2];
    static_assert(NB == 3 && DMA_PER_STAGE == 6, "the slot arithmetic and the counted waits below");
#define PIN_ACC16() do { _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) asm volatile("" : "+a"(acc[c_][b_])); } while (0)
#if defined(LEMON_GEMM_ABLATE) && (LEMON_GEMM_ABLATE & 4)
    constexpr bool MF_ALL = false;      // (diagnostic: the hi x hi product only)
#else
    constexpr bool MF_ALL = true;
#endif
#define MF_BLOCK(c)                                                                                              \
    do {                                                                                                         \
        _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) mfma16(acc[c][b_], wf[c][0], af[b_][0]);               \
        /* W_hi 2^-11 (exact) is made HERE, four MFMAs ahead of its first use, and pinned to this point of the asm   \
           sequence: the MFMAs are asm statements, so hipcc pads no wait states between a v_pk_mul_f16 and an MFMA   \
           that reads its result -- placed right in front of it (where hipcc sinks the multiply if left alone) the   \
           MFMA read the register before the multiply had written it (wrong his products in the first activation     \
           block, seen on the GPU; tests/test_build_guard.py now checks the distance on the generated ISA) */        \
        h16x8 ws_ = wf[c][0] * (_Float16)0.00048828125f;                                                         \
        asm volatile("" : "+v"(ws_));                                                                            \
        if (MF_ALL) { _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) mfma16(acc[c][b_], wf[c][1], af[b_][0]); }  \
        if (MF_ALL) { _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) mfma16(acc[c][b_], ws_, af[b_][1]); }       \
    } while (0)
    int p0 = 0, p1 = 1;                                     // ring positions of slots t, t+1
    PIN_ACC16();
#ifdef LEMON_GEMM_PHASES
    // inside the loop: (a) the DMA wait + barrier B, (b) fragment reads + weight block 0 + barrier B', (c) the other seven
    // weight blocks with the DMA issue.  (s_memtime returns through LGKM_CNT: stamped only where no LDS read is in flight.)
    unsigned long long lp_a = 0, lp_b = 0, lp_c = 0, lt = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lt));
#define LSTAMP(acc) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(now_)); acc += now_ - lt; lt = now_; } while (0)
#else
#define LSTAMP(acc) do { } while (0)
#endif
    for (int t = 0; t < KS; t += 2) {
        PIN_ACC16();
        LSTAMP(lp_c);
        // slots t and t+1 have landed (this wave's share; the barrier makes it everyone's): the only younger DMAs are slot t+2's
#if defined(LEMON_GEMM_ABLATE) && (LEMON_GEMM_ABLATE & 8)
        if (t + 2 < KS) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (t + 2 < KS) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_s_barrier();
#ifdef LEMON_GEMM_PHASES
        if (t == 0) ph1 = __builtin_amdgcn_s_memtime();
#endif
        LSTAMP(lp_a);
        const unsigned sb = (unsigned)((upper ? p1 : p0) * STAGE);
        const unsigned pa = fa + sb, pw = fw + sb;
#if defined(LEMON_GEMM_ABLATE) && (LEMON_GEMM_ABLATE & 2)
        if (t == 0) {   // (diagnostic: fragments read once, every later step reuses them)
#endif
        // LGKM_CNT is a 4-bit counter: never more than 15 LDS reads in flight, or the counted waits below read a wrapped count
        // (seen: 24 reads issued at once -> `lgkmcnt(14)` fell through with the fragments still on their way).  Two batches:
        // 14 reads (activations + weight blocks 0-2), wait for the first ten, then the other ten behind at most four.
#pragma unroll
        for (int b = 0; b < 4; ++b) af[b][0] = lds128(pa, (b >> 1) * 1024 + (b & 1) * 256);
#pragma unroll
        for (int b = 0; b < 4; ++b) af[b][1] = lds128(pa, BLKA + (b >> 1) * 1024 + (b & 1) * 256);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            wf[c][0] = lds128(pw, (c >> 1) * 1024 + (c & 1) * 256);
            wf[c][1] = lds128(pw, BLKW + (c >> 1) * 1024 + (c & 1) * 256);
        }
        // the activation fragments and weight block 0 (LDS returns in order).  As in the 32x32 kernel the waits name the
        // registers they guard, so that no compiler-scheduled use can move in front of them.
        asm volatile("s_waitcnt lgkmcnt(4)"
                     : "+v"(af[0][0]), "+v"(af[1][0]), "+v"(af[2][0]), "+v"(af[3][0]), "+v"(af[0][1]), "+v"(af[1][1]),
                       "+v"(af[2][1]), "+v"(af[3][1]), "+v"(wf[0][0]), "+v"(wf[0][1]) : : "memory");
#pragma unroll
        for (int c = 3; c < 8; ++c) {
            wf[c][0] = lds128(pw, (c >> 1) * 1024 + (c & 1) * 256);
            wf[c][1] = lds128(pw, BLKW + (c >> 1) * 1024 + (c & 1) * 256);
        }
#if defined(LEMON_GEMM_ABLATE) && (LEMON_GEMM_ABLATE & 2)
        }
#endif
        MF_BLOCK(0);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[2][0]), "+v"(wf[2][1]), "+v"(wf[3][0]), "+v"(wf[3][1]), "+v"(wf[4][0]),
                       "+v"(wf[4][1]), "+v"(wf[5][0]), "+v"(wf[5][1]), "+v"(wf[6][0]), "+v"(wf[6][1]), "+v"(wf[7][0]), "+v"(wf[7][1]) : : "memory");
        __builtin_amdgcn_s_barrier();                       // B': every wave has read both slots
        LSTAMP(lp_b);
#ifdef LEMON_GEMM_DMA_BURST
        if (t + 3 < KS) issue(t + 3, p0);
        if (t + 4 < KS) issue(t + 4, p1);
        MF_BLOCK(1); MF_BLOCK(2); MF_BLOCK(3); MF_BLOCK(4); MF_BLOCK(5); MF_BLOCK(6); MF_BLOCK(7);
#else
        // the twelve 1-KB DMA pieces of slots t+3, t+4 go out two at a time between the weight blocks' MFMAs (same order, same
        // counts for the waits above): issued in one burst behind B' they kept the wave from its MFMAs for their whole issue time
#if defined(LEMON_GEMM_ABLATE) && (LEMON_GEMM_ABLATE & 1)
        const bool d3 = false, d4 = false;     // (diagnostic: no operand traffic inside the loop)
#else
        const bool d3 = t + 3 < KS, d4 = t + 4 < KS;
#endif
        const char *as3 = a_src + (size_t)(t + 3) * 2 * BLKA, *ws3 = w_src + (size_t)(t + 3) * 2 * BLKW;
        const char *as4 = a_src + (size_t)(t + 4) * 2 * BLKA, *ws4 = w_src + (size_t)(t + 4) * 2 * BLKW;
        const unsigned la3 = lds0 + p0 * STAGE + wave * (BLKA / 2), lw3 = lds0 + p0 * STAGE + 2 * BLKA + wave * (BLKW / 2);
        const unsigned la4 = lds0 + p1 * STAGE + wave * (BLKA / 2), lw4 = lds0 + p1 * STAGE + 2 * BLKA + wave * (BLKW / 2);
        static_assert(BLKA / 2048 == 2, "two activation pieces per wave and slot");
#if defined(LEMON_GEMM_ABLATE) && (LEMON_GEMM_ABLATE & 8)
        // (diagnostic: the activation pieces -- a third of the operand bytes -- are not fetched: what a third less delivery is worth)
        MF_BLOCK(1);
#else
        MF_BLOCK(1); if (d3) { dma1k(as3, va, la3); dma1k(as3, va + 1024, la3 + 1024); }
#endif
        MF_BLOCK(2); if (d3) { dma1k(ws3, vw, lw3); dma1k(ws3, vw + 1024, lw3 + 1024); }
        MF_BLOCK(3); if (d3) { dma1k(ws3, vw + 2048, lw3 + 2048); dma1k(ws3, vw + 3072, lw3 + 3072); }
#if defined(LEMON_GEMM_ABLATE) && (LEMON_GEMM_ABLATE & 8)
        MF_BLOCK(4);
#else
        MF_BLOCK(4); if (d4) { dma1k(as4, va, la4); dma1k(as4, va + 1024, la4 + 1024); }
#endif
        MF_BLOCK(5); if (d4) { dma1k(ws4, vw, lw4); dma1k(ws4, vw + 1024, lw4 + 1024); }
        MF_BLOCK(6); if (d4) { dma1k(ws4, vw + 2048, lw4 + 2048); dma1k(ws4, vw + 3072, lw4 + 3072); }
        MF_BLOCK(7);
#endif
        p0 = p0 == 0 ? 2 : p0 - 1;                          // (p + 2) mod 3
        p1 = p1 == 0 ? 2 : p1 - 1;
    }
#undef MF_BLOCK
#undef PIN_ACC16
    LSTAMP(lp_c);
#undef LSTAMP
    // ---- epilogue: the lane holds row m = r16 of activation block b and columns n = 4 kg + e of weight block c ----
    const int N = p.n;
#ifdef LEMON_GEMM_PHASES
    const unsigned long long ph2 = __builtin_amdgcn_s_memtime();
    struct PhaseEnd {
        unsigned long long *dbg, a, b, c, r, la, lb, lc; int tid;
        __device__ ~PhaseEnd() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long d = __builtin_amdgcn_s_memtime(), rd = __builtin_amdgcn_s_memrealtime();
            if (dbg && tid == 0) {
                atomicAdd(dbg, b - a); atomicAdd(dbg + 1, c - b); atomicAdd(dbg + 2, d - c); atomicAdd(dbg + 3, 1ull);
                atomicAdd(dbg + 4, rd - r); atomicAdd(dbg + 5, la); atomicAdd(dbg + 6, lb); atomicAdd(dbg + 7, lc);
            }
        }
    } phase_end{p.dbg, ph0, ph1, ph2, rt0, lp_a, lp_b, lp_c, tid};
#endif
    if (EPI == 0 && !FOLD && !EMIT) {
        // fp32 row-major (+ residual) in full 128-byte lines: 2 x 2 accumulator tiles (32 m x 32 n) per pass through the
        // wave-private LDS patch (the ring is free: every wave's reads ended before the last B').  The residual of pass i + 1 is
        // fetched while pass i makes its LDS round trip and stores: fetched inside the pass that uses it, every pass waited a
        // global-load latency (eight per tile: +8.5 % on the output projection, +2.9 % on fc2, tools/micro/gemm_ab.hip).  Rows
        // beyond m read the last valid row (never stored), so that the loads need no per-lane branches.
        // Two patches per wave: pass i + 1's accumulator tiles are written before pass i is read back, so that a pass does not
        // wait for its own LDS round trip; the four bias vectors of the wave's columns are fetched once, up front.
        float *patch0 = reinterpret_cast<float *>(smem) + wave * (2 * 32 * 36);
        const int prow = lane >> 3, pcol = 4 * (lane & 7);
        const bool has_res = p.residual != nullptr;
        const int64_t m_last = p.m - 1;
        float4 bias4[4];
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            bias4[cp] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bias4[cp] = *reinterpret_cast<const float4 *>(p.bias + nt * TN + wn * 128 + cp * 32 + pcol);
        }
        // Addresses.  A full tile (all but the last tile row) needs no row clamp: a wave-uniform 64-bit base (the wave's first
        // row and column) + a 32-bit lane offset + compile-time multiples of N -- hipcc then issues `global_load/store ... s[base]`
        // with one v_add per access; with the clamp every access cost a 64-bit compare-select-multiply chain.
        const bool full = (int64_t)(mt + 1) * TM <= p.m;
        const int64_t wave_off = ((int64_t)mt * TM + wm * (IB * 32)) * N + nt * TN + wn * 128;      // (wave-uniform)
        const char *res_w = reinterpret_cast<const char *>(p.residual + wave_off);
        char *out_w = reinterpret_cast<char *>(reinterpret_cast<float *>(p.out) + wave_off);
        const unsigned lane_off = (unsigned)(prow * N + pcol) * 4u;                                  // (bytes)
        auto load_res = [&](int pass, float4 (&r)[4]) {
            const int n = nt * TN + wn * 128 + (pass >> 1) * 32 + pcol;
            const int64_t m0 = (int64_t)mt * TM + wm * (IB * 32) + (pass & 1) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t m = m0 + 8 * q + prow;
                if (full) r[q] = *reinterpret_cast<const float4 *>(res_w + (lane_off + 4u * (unsigned)(((pass & 1) * 32 + 8 * q) * N + (pass >> 1) * 32)));
                else r[q] = *reinterpret_cast<const float4 *>(p.residual + (m < m_last ? m : m_last) * N + n);
            }
        };
        float4 rn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rn[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_res) load_res(0, rn);
#define PATCH_WRITE(pass_)                                                                                       \
        do {                                                                                                     \
            float *pw_ = patch0 + ((pass_) & 1) * (32 * 36);                                                     \
            _Pragma("unroll") for (int ci = 0; ci < 2; ++ci) _Pragma("unroll") for (int bi = 0; bi < 2; ++bi) { \
                const f32x4 a = acc[2 * ((pass_) >> 1) + ci][2 * ((pass_) & 1) + bi];                            \
                *reinterpret_cast<float4 *>(pw_ + (bi * 16 + r16) * 36 + ci * 16 + 4 * kg) = make_float4(a[0], a[1], a[2], a[3]); \
            }                                                                                                    \
        } while (0)
        // The eight passes, in two forms: tiles that lie wholly inside the m rows (all but the last tile row) store without
        // per-lane conditions -- behind `if (m < p.m)` hipcc wraps every load / store in its own exec-masked branch and waits
        // vmcnt(0) / lgkmcnt(0) at each of them (four serialised LDS reads and a full drain of the previous pass's stores per
        // pass: the fp32 epilogue was 20 % of a k = 768 tile's cycles) --, the ragged tile row keeps the conditions.
#define EPI0_PASSES(MASKED)                                                                                      \
        _Pragma("unroll") for (int pass = 0; pass < 8; ++pass) {                                                 \
            const int cp = pass >> 1, bp = pass & 1;                                                             \
            const int n = nt * TN + wn * 128 + cp * 32 + pcol;                                                   \
            const float4 bv = bias4[cp];                                                                         \
            const int64_t m0 = (int64_t)mt * TM + wm * (IB * 32) + bp * 32;                                      \
            float4 r[4];                                                                                         \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) r[q] = rn[q];                                          \
            if (has_res && pass + 1 < 8) load_res(pass + 1, rn);                                                 \
            if (pass + 1 < 8) PATCH_WRITE(pass + 1);                                                             \
            const float *pr = patch0 + (pass & 1) * (32 * 36);                                                   \
            float4 v[4];                                                                                         \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(pr + (8 * q + prow) * 36 + pcol); \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
                const int64_t m = m0 + 8 * q + prow;                                                             \
                const float4 o4_ = make_float4(p.alpha * v[q].x + bv.x + r[q].x, p.alpha * v[q].y + bv.y + r[q].y, p.alpha * v[q].z + bv.z + r[q].z, p.alpha * v[q].w + bv.w + r[q].w); \
                if (!(MASKED)) *reinterpret_cast<float4 *>(out_w + (lane_off + 4u * (unsigned)((bp * 32 + 8 * q) * N + cp * 32))) = o4_; \
                else if (m < p.m) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.out) + m * N + n) = o4_; \
            }                                                                                                    \
        }
        PATCH_WRITE(0);
        if (full) { EPI0_PASSES(false) } else { EPI0_PASSES(true) }
#undef EPI0_PASSES
#undef PATCH_WRITE
        return;
    }
    if (EPI == 0) {
        // The fp32 epilogue with a folded LayerNorm on one side or the other (see "LayerNorm fold" at lemon_linear_f16x3t_ln).  As
        // above -- 32 x 32 patches through LDS, the next pass's residual, bias and weight-row sums fetched one pass ahead --, with
        // the passes ordered row half first (all four column groups of rows 0-31, then of rows 32-63): a row's statistics are
        // complete after four passes and only four rows' running sums are live at a time.
        //   FOLD: out = rstd_m (alpha acc) - mean_m rstd_m colsum_n + bias_n (+ residual); the wave's 64 (rstd, -mean rstd) pairs
        //         wait in LDS
        //   EMIT: the stored value also goes out as the next GEMM's tile-major activation operand (eight rows x 16 B = one
        //         128-byte line per store instruction and k chunk), and as a (mean, M2) partial per row and 128-column group:
        //         shifted sums over the lane's 16 values (shift = the first one: the sums stay at the size of the row's spread),
        //         equal-count pairwise merges over the row's eight lanes
        // Addresses as in the plain epilogue: full tiles use a wave-uniform base + 32-bit lane offset + compile-time terms (for the
        // tile-major operand too: split3.hpp tiled_off taken apart into its wave, lane and (pass, q) parts).
        float *patch0 = reinterpret_cast<float *>(smem) + wave * (2 * 32 * 36);
        float2 *saff = reinterpret_cast<float2 *>(smem + 4 * (2 * 32 * 36) * 4) + wave * 64;
        const int prow = lane >> 3, pcol = 4 * (lane & 7);
        const bool has_res = p.residual != nullptr;
        // EMIT only: the residual may arrive as a tile-major operand (what the EMIT GEMM in front of this one left: the same tile
        // offsets as this GEMM's own operand output), and the fp32 result may be left out (out = null) when every consumer reads the
        // operand form -- the output projection of a block in the chain then writes 6 instead of 10 bytes per element
        constexpr bool res_t = (LEAN & 2) != 0;
        constexpr bool has_out = (LEAN & 1) == 0;
        const int64_t m_last = p.m - 1;
        const int64_t mw = (int64_t)mt * TM + wm * (IB * 32);          // the wave's first row
        const bool full = (int64_t)(mt + 1) * TM <= p.m;
        const int n_w = nt * TN + wn * 128;                            // the wave's first column
        const int64_t wave_off = mw * N + n_w;
        const char *res_w = reinterpret_cast<const char *>(p.residual + wave_off);
        char *out_w = reinterpret_cast<char *>(reinterpret_cast<float *>(p.out) + wave_off);
        const unsigned lane_off = (unsigned)(prow * N + pcol) * 4u;
        // tile-major operand (halves): ((mt (N / 16) + n / 16) 2 + part) (TM 16) + (r / 32) 512 + ((n / 8) & 1) 256 + (r % 32) 8 + n % 8
        const int64_t emit_off = 2 * ((((int64_t)mt * (N >> 4) + (n_w >> 4)) * 2) * (TM * 16) + wm * 2 * 512);
        char *emit_w = reinterpret_cast<char *>(p.emit_t) + emit_off;
        const char *rest_w = reinterpret_cast<const char *>(p.residual_t) + emit_off;
        const unsigned emit_lane = 2u * (unsigned)(((lane & 7) >> 2) * 2 * (TM * 16) + (((lane & 7) >> 1) & 1) * 256 + prow * 8 + 4 * (lane & 1));
        if (FOLD) { const int64_t m = mw + lane; saff[lane] = p.row_aff[m < m_last ? m : m_last]; }
        auto load_next = [&](int pass, float4 (&r)[4], float4 &bv, float4 &cs) {
            const int cpn = pass & 3, bpn = pass >> 2;
            const int n = n_w + cpn * 32 + pcol;
            bv = make_float4(0.f, 0.f, 0.f, 0.f); cs = bv;
            if (p.bias) bv = *reinterpret_cast<const float4 *>(p.bias + n);
            if (FOLD) cs = *reinterpret_cast<const float4 *>(p.colsum + n);
            if (res_t) {
                // (hi in .xy, lo in .zw of the same four registers; rows beyond m are the operand's pad rows: never stored)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const char *a_ = rest_w + (emit_lane + (unsigned)(cpn * 16384 + bpn * 1024 + q * 128));
                    const float2 h2_ = *reinterpret_cast<const float2 *>(a_), l2_ = *reinterpret_cast<const float2 *>(a_ + TM * 16 * 2);
                    r[q] = make_float4(h2_.x, h2_.y, l2_.x, l2_.y);
                }
            } else if (has_res) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t m = mw + bpn * 32 + 8 * q + prow;
                    if (full) r[q] = *reinterpret_cast<const float4 *>(res_w + (lane_off + 4u * (unsigned)((bpn * 32 + 8 * q) * N + cpn * 32)));
                    else r[q] = *reinterpret_cast<const float4 *>(p.residual + (m < m_last ? m : m_last) * N + n);
                }
            }
        };
        float4 rn[4], bn, cn;
#pragma unroll
        for (int q = 0; q < 4; ++q) rn[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        load_next(0, rn, bn, cn);
        float st_s[4], st_1[4], st_2[4];
#define PATCH_WRITE(pass_)                                                                                       \
        do {                                                                                                     \
            float *pw_ = patch0 + ((pass_) & 1) * (32 * 36);                                                     \
            _Pragma("unroll") for (int ci = 0; ci < 2; ++ci) _Pragma("unroll") for (int bi = 0; bi < 2; ++bi) { \
                const f32x4 a = acc[2 * ((pass_) & 3) + ci][2 * ((pass_) >> 2) + bi];                            \
                *reinterpret_cast<float4 *>(pw_ + (bi * 16 + r16) * 36 + ci * 16 + 4 * kg) = make_float4(a[0], a[1], a[2], a[3]); \
            }                                                                                                    \
        } while (0)
#define EPI0_PASSES(MASKED)                                                                                      \
        _Pragma("unroll") for (int pass = 0; pass < 8; ++pass) {                                                 \
            const int cp = pass & 3, bp = pass >> 2;                                                             \
            const int n = n_w + cp * 32 + pcol;                                                                  \
            const int64_t m0 = mw + bp * 32;                                                                     \
            float4 r[4];                                                                                         \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) r[q] = rn[q];                                          \
            const float4 bv = bn, cs = cn;                                                                       \
            if (pass + 1 < 8) load_next(pass + 1, rn, bn, cn);                                                   \
            if (pass + 1 < 8) PATCH_WRITE(pass + 1);                                                             \
            const float *pr = patch0 + (pass & 1) * (32 * 36);                                                   \
            float4 v[4];                                                                                         \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(pr + (8 * q + prow) * 36 + pcol); \
            float2 af[4];                                                                                        \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) af[q] = FOLD ? saff[bp * 32 + 8 * q + prow] : make_float2(1.0f, 0.0f); \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
                const int64_t m = m0 + 8 * q + prow;                                                             \
                float o_[4] = {p.alpha * v[q].x, p.alpha * v[q].y, p.alpha * v[q].z, p.alpha * v[q].w};          \
                if (FOLD) {                                                                                      \
                    /* the four products are kept as single registers: left alone hipcc pairs them into                \
                       `v_pk_mul_f32 ... op_sel:[0,1]` (low lane = HIGH dword of the (rstd, -mean rstd) pair), and with  \
                       that instruction right behind the pair's arrival the low results of lanes 48-63 came out wrong  \
                       now and then (seen on the GPU, never with the four scalar multiplies; tests/test_build_guard.py   \
                       refuses the pattern in these kernels) */                                                       \
                    float t_[4] = {af[q].y * cs.x, af[q].y * cs.y, af[q].y * cs.z, af[q].y * cs.w};              \
                    asm volatile("" : "+v"(t_[0]), "+v"(t_[1]), "+v"(t_[2]), "+v"(t_[3]));                       \
                    o_[0] = af[q].x * o_[0] + t_[0]; o_[1] = af[q].x * o_[1] + t_[1];                             \
                    o_[2] = af[q].x * o_[2] + t_[2]; o_[3] = af[q].x * o_[3] + t_[3];                             \
                }                                                                                                \
                float rr_[4] = {r[q].x, r[q].y, r[q].z, r[q].w};                                                 \
                if (res_t) {   /* hi + lo 2^-11, each term exact in fp32, one fused step (v_fma_mix_f32) */            \
                    const h16x4 rh_ = __builtin_bit_cast(h16x4, make_float2(r[q].x, r[q].y)), rl_ = __builtin_bit_cast(h16x4, make_float2(r[q].z, r[q].w)); \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) rr_[e] = __builtin_fmaf((float)rl_[e], 0.00048828125f, (float)rh_[e]); \
                }                                                                                                \
                o_[0] += bv.x + rr_[0]; o_[1] += bv.y + rr_[1]; o_[2] += bv.z + rr_[2]; o_[3] += bv.w + rr_[3];   \
                const float4 o4_ = make_float4(o_[0], o_[1], o_[2], o_[3]);                                      \
                if (has_out) {                                                                                   \
                    if (!(MASKED)) *reinterpret_cast<float4 *>(out_w + (lane_off + 4u * (unsigned)((bp * 32 + 8 * q) * N + cp * 32))) = o4_; \
                    else if (m < p.m) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.out) + m * N + n) = o4_; \
                }                                                                                                \
                if (EMIT) {                                                                                      \
                    /* (rows beyond m land in the operand's pad rows: never read into a stored result)                \
                       lo 2^11 = o 2^11 - hi 2^11 exactly, in one fused step from the fp16 value (v_fma_mix_f32) */    \
                    h16x4 hi_, lo_;                                                                              \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                              \
                        const _Float16 hh_ = (_Float16)o_[e];                                                    \
                        hi_[e] = hh_; lo_[e] = (_Float16)__builtin_fmaf((float)hh_, -2048.0f, o_[e] * 2048.0f);  \
                    }                                                                                            \
                    char *base_ = emit_w + (emit_lane + (unsigned)(cp * 16384 + bp * 1024 + q * 128));           \
                    *reinterpret_cast<h16x4 *>(base_) = hi_;                                                     \
                    *reinterpret_cast<h16x4 *>(base_ + TM * 16 * 2) = lo_;                                       \
                    if (cp == 0) {                                                                               \
                        float s0_ = o_[0];                /* (its own register: the shift is never the odd half of a pair */ \
                        asm volatile("" : "+v"(s0_));     /*  a packed op would have to reach with op_sel, see FOLD) */  \
                        st_s[q] = s0_; st_1[q] = 0.0f; st_2[q] = 0.0f;                                           \
                    }                                                                                            \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                              \
                        const float d_ = o_[e] - st_s[q];                                                        \
                        st_1[q] += d_; st_2[q] += d_ * d_;                                                       \
                    }                                                                                            \
                    if (cp == 3) {                                                                               \
                        float mean = st_s[q] + st_1[q] * 0.0625f;                                                \
                        float m2 = st_2[q] - st_1[q] * st_1[q] * 0.0625f;                                        \
                        float cnt = 16.0f;                                                                       \
                        _Pragma("unroll") for (int off = 1; off < 8; off <<= 1) {                                \
                            const float mo = __shfl_xor(mean, off), m2o = __shfl_xor(m2, off);                   \
                            const float dm = mo - mean;                                                          \
                            m2 = m2 + m2o + dm * dm * (0.5f * cnt);                                              \
                            mean = 0.5f * (mean + mo);                                                           \
                            cnt *= 2.0f;                                                                         \
                        }                                                                                        \
                        if ((lane & 7) == 0 && m < p.m)                                                          \
                            *reinterpret_cast<float2 *>(p.emit_stats + (m * (2 * p.n_tiles) + 2 * nt + wn) * 2) = make_float2(mean, m2); \
                    }                                                                                            \
                }                                                                                                \
            }                                                                                                    \
        }
        PATCH_WRITE(0);
        if (full) { EPI0_PASSES(false) } else { EPI0_PASSES(true) }
#undef EPI0_PASSES
#undef PATCH_WRITE
        return;
    }
    float4 bias8[8];                                        // (fetched once: behind the operand stores below hipcc must reload them)
    float4 cs8[8];                                          // FOLD: the weight-row sums of the lane's columns
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        bias8[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bias8[c] = *reinterpret_cast<const float4 *>(p.bias + nt * TN + wn * 128 + c * 16 + 4 * kg);
        if (FOLD) cs8[c] = *reinterpret_cast<const float4 *>(p.colsum + nt * TN + wn * 128 + c * 16 + 4 * kg);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t m = (int64_t)mt * TM + wm * (IB * 32) + b * 16 + r16;
        float2 af = make_float2(1.0f, 0.0f);
        if (FOLD) af = p.row_aff[m < p.m - 1 ? m : p.m - 1];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int n = nt * TN + wn * 128 + c * 16 + 4 * kg;
            const float4 bv = bias8[c];
            float o[4] = {p.alpha * acc[c][b][0], p.alpha * acc[c][b][1], p.alpha * acc[c][b][2], p.alpha * acc[c][b][3]};
            if (FOLD) {
                const float4 cs = cs8[c];
                float t[4] = {af.y * cs.x, af.y * cs.y, af.y * cs.z, af.y * cs.w};     // (single registers: see the fp32 epilogue)
                asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
                o[0] = af.x * o[0] + t[0]; o[1] = af.x * o[1] + t[1]; o[2] = af.x * o[2] + t[2]; o[3] = af.x * o[3] + t[3];
            }
            o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
            // the next GEMM's activation operand (its k = this n): the lane's four values are half a 16-byte slot, a wave's
            // store instruction covers two runs of 256 contiguous bytes
            h16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = epilogue_act<EPI>(o[e]);
                unsigned short a_, b_, c_;
                split2h<false>(v, a_, b_, c_);
                hi[e] = __builtin_bit_cast(_Float16, a_); lo[e] = __builtin_bit_cast(_Float16, c_);
            }
            unsigned short *base = reinterpret_cast<unsigned short *>(p.out) + tiled_off(TM, m, n, 0, N);
            *reinterpret_cast<h16x4 *>(base) = hi;
            *reinterpret_cast<h16x4 *>(base + TM * 16) = lo;
        }
    }
}

// fp32 [n, k] weight -> tile-major fp16 pairs of w * wscale (one thread per 8 consecutive k)
__global__ __launch_bounds__(256) void k_pack_weight_t(const float *__restrict__ w, int n, int k, float wscale, unsigned short *__restrict__ wt) {
    const int nch = k >> 3;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * nch) return;
    const int64_t r = t / nch;
    const int c = (int)(t - r * nch);
    const float4 *src = reinterpret_cast<const float4 *>(w + r * (int64_t)k + 8 * c);
    store_tiled8<TILE_W_ROWS, true>(wt, r, k, c, src[0], src[1], wscale);
}

// tile-major activation operand -> fp32 [rows, k] (hi + lo 2^-11): test / debugging aid
__global__ __launch_bounds__(256) void k_unpack_act_t(const unsigned short *__restrict__ at, int64_t rows, int k, float *__restrict__ y) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * k) return;
    const int64_t r = t / k;
    const int c = (int)(t - r * k);
    const float hi = f16_val(at[tiled_off(TILE_A_ROWS, r, c, 0, k)]), lo = f16_val(at[tiled_off(TILE_A_ROWS, r, c, 1, k)]);
    y[t] = hi + lo * 0.00048828125f;
}

// a row's (mean, M2) partials over `per`-column groups -> (rstd, -mean rstd) of its LayerNorm (one thread per row)
__global__ __launch_bounds__(256) void k_ln_finalize(const float *__restrict__ part, int64_t rows, int np, float per, float eps, float2 *__restrict__ aff) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float2 *pr = reinterpret_cast<const float2 *>(part) + r * np;
    float mean = 0.0f;
    for (int i = 0; i < np; ++i) mean += pr[i].x;
    mean /= (float)np;
    float m2 = 0.0f;
    for (int i = 0; i < np; ++i) { const float d = pr[i].x - mean; m2 += pr[i].y + per * d * d; }
    const float rstd = rsqrtf(m2 / (per * (float)np) + eps);
    const bool far = fabsf(mean) * rstd > LEMON_LN_FOLD_MAX_SHIFT;           // (see common.hpp: the caller falls back)
    aff[r] = far ? make_float2(__builtin_nanf(""), __builtin_nanf("")) : make_float2(rstd, -mean * rstd);
}

// ---- host-side state of the launcher ----
constexpr int MAX_DEVICES = 64;
std::mutex g_attr_mu;
bool g_attr_set[MAX_DEVICES] = {};
// optional timing of every launch (lemon_linear_f16x3t_set_profiling): HIP events on the launch stream from a pool that GROWS with
// the launches it has to bracket (round 4 had a fixed pool of 8 192 pairs and silently stopped recording behind it: a 20-step
// bench region has 18 200 launches), read back (and the pool rewound) by lemon_linear_f16x3t_profile_read -- bench.py's roofline
// object for the step's dominant kernel.  A launch that cannot get its events fails (LEMON_E_HIP): nothing is dropped.
struct GemmProf {
    bool on = false;
    std::vector<hipEvent_t> ev;        // pairs
    std::vector<double> flops;
    size_t used = 0;                   // events handed out
} g_prof;
constexpr size_t PROF_POOL = 2 * 1024; // events created when profiling is switched on; more are made as launches need them
int g_gm = 0, g_gn = 0;                // tile-walk override (tools/micro); 0: the defaults
#ifdef LEMON_GEMM_PHASES
unsigned long long *g_phase_dbg = nullptr;
#endif
int g_mfma_shape = 0;                  // 0: not decided yet ($LEMON_GEMM_MFMA, default 16); tools/micro sets it directly
void walk_override() {
    static bool read = false;
    if (read) return;
    read = true;
    const char *e = getenv("LEMON_GEMM_WALK");
    int a = 0, b = 0;
    if (e && sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0 && a <= 1024 && b <= 64) { g_gm = a; g_gn = b; }
}
int mfma_shape() {
    if (g_mfma_shape == 0) {
        const char *e = getenv("LEMON_GEMM_MFMA");
        g_mfma_shape = (e && atoi(e) == 32) ? 32 : 16;
    }
    return g_mfma_shape;
}

}  // namespace

extern "C" int lemon_pack_weight_f16x3t(const float *w_dev, int n, int k, float wscale, uint16_t *wt_dev, void *stream_) {
    LEMON_REQUIRE(n > 0 && k > 0 && n % TN == 0 && k % 16 == 0, "n a multiple of 256, k a multiple of 16");
    LEMON_REQUIRE(w_dev && wt_dev && ((((uintptr_t)w_dev) | ((uintptr_t)wt_dev)) & 15) == 0, "aligned pointers");
    int e = 0;
    LEMON_REQUIRE(wscale > 0.0f && std::frexp(wscale, &e) == 0.5f, "wscale must be a power of two");
    const int64_t threads = (int64_t)n * (k >> 3);
    hipLaunchKernelGGL(k_pack_weight_t, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w_dev, n, k, wscale, wt_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_unpack_act_f16x3t(const uint16_t *at_dev, int64_t rows, int k, float *y_dev, void *stream_) {
    LEMON_REQUIRE(rows >= 0 && k > 0 && k % 16 == 0, "rows >= 0, k a multiple of 16");
    if (rows == 0) return LEMON_OK;
    LEMON_REQUIRE(at_dev && y_dev, "null pointer");
    const int64_t threads = rows * k;
    hipLaunchKernelGGL(k_unpack_act_t, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, at_dev, rows, k, y_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

// ---- LayerNorm fold ---------------------------------------------------------------------------------------------------------
// LN(x) W^T + b = rstd (x W'^T - mean c) + b' with W' = W diag(gamma), c = W' 1, b' = b + W beta: the GEMM in front of which a
// LayerNorm stands can take the RAW residual stream x as its activation operand and apply the row's (rstd, -mean rstd) and the
// weight-row sums in its epilogue (FOLD) -- if the GEMM that PRODUCES x (output projection, fc2: fp32 + residual) also writes x
// as the tile-major operand and leaves per-row statistics (EMIT): the LayerNorm pass (one read + one write of the token matrix,
// 4.7 % of the headline step) disappears.  Statistics are (mean, M2) partials per row and 128-column group, shifted sums inside
// a lane and pairwise merges (no E[x^2] - mean^2 cancellation); k_ln_finalize merges a row's partials.  Error: the split
// operand now carries x instead of LN(x), so a product term's rounding (2^-22) scales with |x_k| rstd instead of |x_k - mean|
// rstd -- a factor sqrt(1 + mean^2 / var) on the GEMM's error, 1.0x for the zero-mean rows of a transformer's residual stream.
static int linear_f16x3t_impl(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                              int64_t m, int n, int k, float alpha, int act, int out_operand, void *out_dev,
                              const float *row_aff_dev, const float *colsum_dev, uint16_t *emit_t_dev, float *emit_stats_dev, void *stream_,
                              const uint16_t *residual_t_dev = nullptr);

extern "C" int lemon_linear_f16x3t(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                                   int64_t m, int n, int k, float alpha, int act, int out_operand, void *out_dev, void *stream_) {
    return linear_f16x3t_impl(at_dev, wt_dev, bias_dev, residual_dev, m, n, k, alpha, act, out_operand, out_dev, nullptr, nullptr, nullptr, nullptr, stream_);
}

extern "C" int lemon_linear_f16x3t_ln(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                                      int64_t m, int n, int k, float alpha, int act, int out_operand, void *out_dev,
                                      const float *row_aff_dev, const float *colsum_dev, uint16_t *emit_t_dev, float *emit_stats_dev, void *stream_) {
    LEMON_REQUIRE((row_aff_dev != nullptr) == (colsum_dev != nullptr), "row_aff and colsum come together");
    LEMON_REQUIRE((emit_t_dev != nullptr) == (emit_stats_dev != nullptr), "emit_t and emit_stats come together");
    LEMON_REQUIRE(!emit_t_dev || (!out_operand && act == LEMON_ACT_NONE), "the operand + statistics output rides on the fp32 form");
    LEMON_REQUIRE(!(row_aff_dev && emit_t_dev), "a GEMM either consumes a folded LayerNorm or produces the next one's input");
    LEMON_REQUIRE(!(row_aff_dev || emit_t_dev) || k % 32 == 0, "the LayerNorm fold needs k a multiple of 32 (16x16x32 kernel)");
    return linear_f16x3t_impl(at_dev, wt_dev, bias_dev, residual_dev, m, n, k, alpha, act, out_operand, out_dev, row_aff_dev, colsum_dev, emit_t_dev, emit_stats_dev, stream_);
}

// The producing GEMM of a block chain (output projection, fc2) in its leanest form: lemon_linear_f16x3t_ln's EMIT side with
//   residual_t_dev  the residual as the tile-major operand an EMIT GEMM in front left (x = hi + lo 2^-11: 22 significant bits, one
//                   more rounding of the size the split GEMMs make anyway), instead of residual_dev (fp32); at most one of the two
//   out_dev = NULL  no fp32 result: the consumers read emit_t_dev (the next GEMM as its operand, the one after as its residual)
extern "C" int lemon_linear_f16x3t_chain(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                                         const uint16_t *residual_t_dev, int64_t m, int n, int k, float alpha, float *out_dev,
                                         uint16_t *emit_t_dev, float *emit_stats_dev, void *stream_) {
    LEMON_REQUIRE(emit_t_dev && emit_stats_dev, "the chain form always leaves the operand and its row statistics");
    LEMON_REQUIRE(!(residual_dev && residual_t_dev), "one residual: fp32 or operand form");
    LEMON_REQUIRE(k % 32 == 0, "k a multiple of 32 (16x16x32 kernel)");
    LEMON_REQUIRE((((uintptr_t)residual_t_dev) & 15) == 0, "16-byte aligned pointers");
    return linear_f16x3t_impl(at_dev, wt_dev, bias_dev, residual_dev, m, n, k, alpha, LEMON_ACT_NONE, 0, out_dev, nullptr, nullptr, emit_t_dev,
                              emit_stats_dev, stream_, residual_t_dev);
}

static int linear_f16x3t_impl(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                              int64_t m, int n, int k, float alpha, int act, int out_operand, void *out_dev,
                              const float *row_aff_dev, const float *colsum_dev, uint16_t *emit_t_dev, float *emit_stats_dev, void *stream_,
                              const uint16_t *residual_t_dev) {
    LEMON_REQUIRE(m >= 0 && n > 0 && k > 0 && n % TN == 0 && k % 16 == 0, "m >= 0, n a multiple of 256, k a multiple of 16");
    if (m == 0) return LEMON_OK;
    LEMON_REQUIRE(at_dev && wt_dev && (out_dev || emit_t_dev), "null pointer");
    LEMON_REQUIRE(((((uintptr_t)row_aff_dev) | ((uintptr_t)colsum_dev) | ((uintptr_t)emit_t_dev) | ((uintptr_t)emit_stats_dev)) & 15) == 0, "16-byte aligned pointers");
    LEMON_REQUIRE(((((uintptr_t)at_dev) | ((uintptr_t)wt_dev) | ((uintptr_t)out_dev) | ((uintptr_t)bias_dev) | ((uintptr_t)residual_dev)) & 15) == 0,
                  "16-byte aligned pointers");
    LEMON_REQUIRE((act == LEMON_ACT_NONE && !out_operand) || ((act == LEMON_ACT_SILU || act == LEMON_ACT_GELU) && out_operand && !residual_dev),
                  "supported forms: act none -> fp32 [m, n] (+ residual); act SiLU / GELU -> tile-major operand (no residual)");
    LEMON_REQUIRE((m + TM - 1) / TM < ((int64_t)1 << 24), "m < 2^31");
    GemmParams p;
    p.at = reinterpret_cast<const char *>(at_dev); p.wt = reinterpret_cast<const char *>(wt_dev);
    p.bias = bias_dev; p.residual = residual_dev; p.out = out_dev;
    p.row_aff = reinterpret_cast<const float2 *>(row_aff_dev); p.colsum = colsum_dev;
    p.emit_t = reinterpret_cast<unsigned short *>(emit_t_dev); p.emit_stats = emit_stats_dev;
    p.residual_t = reinterpret_cast<const unsigned short *>(residual_t_dev);
    p.m = m; p.n = n; p.ks = k / 16; p.m_tiles = (int)((m + TM - 1) / TM); p.n_tiles = n / TN; p.alpha = alpha;
    // super-block of the tile walk: gn = the largest divisor of the n-tile count up to 4 (a gn that does not divide it leaves
    // every other XCD with half-empty super-blocks: +25 % time measured), gm so that an XCD's 64 resident workgroups cover one
    // or two super-blocks.  Measured against 32 x 1 at the tower shapes (tools/micro/gemm_ab.hip walk): -1 ... -3 %; the
    // operand re-fetches this saves (FETCH_SIZE 5.0 GB for fc1 at 32 x 1: the activations once per weight tile column) are
    // served by the Infinity Cache either way.  LEMON_GEMM_WALK=gm,gn overrides.
    walk_override();
    if (g_gm > 0) { p.gm = g_gm; p.gn = g_gn > 0 ? g_gn : 1; }
    else {
        p.gn = p.n_tiles % 4 == 0 ? 4 : p.n_tiles % 3 == 0 ? 3 : p.n_tiles % 2 == 0 ? 2 : 1;
        p.gm = p.gn == 1 ? GM : p.gn == 3 ? 8 : 16;
    }
    if (p.gn > p.n_tiles) p.gn = p.n_tiles;
    const int64_t blocks = (int64_t)((p.m_tiles + p.gm - 1) / p.gm) * ((p.n_tiles + p.gn - 1) / p.gn);
    int64_t grid = ((blocks + 7) / 8) * p.gm * p.gn * 8;
    LEMON_REQUIRE(grid < ((int64_t)1 << 31), "grid size");
    p.vblocks = (int)grid;
#ifdef LEMON_GEMM_PHASES
    p.dbg = g_phase_dbg;
#endif
    const size_t lds = (size_t)NB * STAGE;
    // the 72 KB of dynamic LDS need the attribute on every DEVICE this process launches on (it is per device, not per process)
    int dev = 0;
    LEMON_HIP_CHECK(hipGetDevice(&dev));
    LEMON_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index");
    {
        std::lock_guard<std::mutex> lock(g_attr_mu);
        if (!g_attr_set[dev]) {
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<0, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<1, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<0, false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<0, false, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<0, false, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3t16<2, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            g_attr_set[dev] = true;
        }
    }
    // matrix instruction: 16x16x32 where the k extent allows whole k32 steps (every tower width does), else 32x32x16;
    // LEMON_GEMM_MFMA=32 forces the latter (A/B runs)
    const bool mf16 = mfma_shape() == 16 && p.ks % 2 == 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_attr_mu);
        if (g_prof.on) {
            while (g_prof.used + 2 > g_prof.ev.size()) {       // every launch of a profiled region is bracketed
                hipEvent_t e = nullptr;
                LEMON_HIP_CHECK(hipEventCreate(&e));
                g_prof.ev.push_back(e);
            }
            ev0 = g_prof.ev[g_prof.used]; ev1 = g_prof.ev[g_prof.used + 1];
            g_prof.used += 2;
            g_prof.flops.push_back(2.0 * (double)m * (double)n * 3.0 * (double)k);      // the kernel's own arithmetic: three fp16 products
        }
    }
    if (ev0) LEMON_HIP_CHECK(hipEventRecord(ev0, (hipStream_t)stream_));
    const bool fold = row_aff_dev != nullptr, emit = emit_t_dev != nullptr;
    const bool gelu = act == LEMON_ACT_GELU;
    if (fold || emit) {
        if (fold && out_operand && gelu) hipLaunchKernelGGL((k_gemm_f16x3t16<2, true, false>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else if (fold && out_operand) hipLaunchKernelGGL((k_gemm_f16x3t16<1, true, false>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else if (fold) hipLaunchKernelGGL((k_gemm_f16x3t16<0, true, false>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else if (!residual_t_dev && out_dev) hipLaunchKernelGGL((k_gemm_f16x3t16<0, false, true>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else if (!residual_t_dev) hipLaunchKernelGGL((k_gemm_f16x3t16<0, false, true, 1>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else if (out_dev) hipLaunchKernelGGL((k_gemm_f16x3t16<0, false, true, 2>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else hipLaunchKernelGGL((k_gemm_f16x3t16<0, false, true, 3>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
    } else if (mf16) {
        if (out_operand && gelu) hipLaunchKernelGGL((k_gemm_f16x3t16<2, false, false>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else if (out_operand) hipLaunchKernelGGL((k_gemm_f16x3t16<1, false, false>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else hipLaunchKernelGGL((k_gemm_f16x3t16<0, false, false>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
    } else {
        if (out_operand && gelu) hipLaunchKernelGGL(k_gemm_f16x3t<2>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else if (out_operand) hipLaunchKernelGGL(k_gemm_f16x3t<1>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
        else hipLaunchKernelGGL(k_gemm_f16x3t<0>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, p);
    }
    LEMON_HIP_CHECK(hipGetLastError());
    if (ev1) LEMON_HIP_CHECK(hipEventRecord(ev1, (hipStream_t)stream_));
    return LEMON_OK;
}

extern "C" int lemon_linear_f16x3t_set_mfma(int shape) {
    LEMON_REQUIRE(shape == 0 || shape == 16 || shape == 32, "shape: 16, 32 or 0 (the default: $LEMON_GEMM_MFMA, else 16)");
    const int prev = mfma_shape();
    g_mfma_shape = shape;
    return prev;
}

extern "C" int lemon_linear_f16x3t_set_profiling(int on) {
    std::lock_guard<std::mutex> lock(g_attr_mu);
    if (on && g_prof.ev.empty()) {
        g_prof.ev.resize(PROF_POOL);
        for (auto &e : g_prof.ev) LEMON_HIP_CHECK(hipEventCreate(&e));
    }
    g_prof.on = on != 0;
    return LEMON_OK;
}

extern "C" int lemon_linear_f16x3t_profile_read(int64_t *launches, double *kernel_ms, double *flops) {
    LEMON_REQUIRE(launches && kernel_ms && flops, "null pointer");
    std::lock_guard<std::mutex> lock(g_attr_mu);
    *launches = (int64_t)(g_prof.used / 2); *kernel_ms = 0.0; *flops = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        LEMON_HIP_CHECK(hipEventSynchronize(g_prof.ev[i + 1]));
        float ms = 0.0f;
        LEMON_HIP_CHECK(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
        *kernel_ms += ms; *flops += g_prof.flops[i / 2];
    }
    g_prof.used = 0; g_prof.flops.clear();
    return LEMON_OK;
}

extern "C" int lemon_ln_finalize(const float *partials_dev, int64_t rows, int width, float eps, float *row_aff_dev, void *stream_) {
    LEMON_REQUIRE(rows >= 0 && width > 0 && width % TN == 0, "rows >= 0, width a multiple of 256");
    if (rows == 0) return LEMON_OK;
    LEMON_REQUIRE(partials_dev && row_aff_dev, "null pointer");
    hipLaunchKernelGGL(k_ln_finalize, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, partials_dev, rows, width / 128, 128.0f, eps,
                       reinterpret_cast<float2 *>(row_aff_dev));
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
