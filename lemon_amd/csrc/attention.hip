// Fused multi-head self-attention for the CLIP towers of the embedding stage (gfx950 only).
//
// Replaces, inside encode_image / encode_text (lib/models/downstream_models.py:30-41 -> HF CLIPAttention;
// in-tree twin lib/models/chexzero_clip.py:191-212 nn.MultiheadAttention), the chain
//   view/permute(qkv) -> softmax(q k^T / sqrt(hd) [+ causal mask]) v -> transpose -> reshape
// with ONE pass: reads the packed projection output qkv[B, L, 3, H, 64] (what the fused QKV GEMM
// writes) and writes out[B, L, H*64] (what the output projection reads).  No [B,H,L,L] matrix, no
// permute copies: algorithmic HBM traffic 16*H*64 B per token (3 reads + 1 write).
//
// One workgroup per (batch, head); wave t owns queries 32t..32t+31.  K and V of the head are staged in
// LDS (row pitch 68 floats: 16-B aligned, conflict-free 128-bit reads).  Per 32-key tile:
//   S^T = K Q^T      32 x v_mfma_f32_32x32x2_f32   A = K rows from LDS, B = the wave's Q rows (registers)
//                    -> lane (i = lane&31, h) holds its OWN query's 16 scores: softmax statistics are
//                       in-lane reductions plus one cross-half shuffle (flash-style running max / sum)
//   O^T += V^T P^T   2 x 16 MFMAs: B = the score registers as they are, A = V columns from LDS
// fp32 throughout (the reference runs the encoders in fp32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <algorithm>

#include "common.hpp"
#include "split3.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;        // head dimension (every CLIP variant the reference loads: 768/12, 512/8, 1024/16)
constexpr int PITCH = 68;     // LDS row pitch in floats

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// eight fp32 values -> fp16 hi and lo * 2^11 (split3.hpp: split2h): hi + lo 2^-11 carries 22 bits + sign of each value
__device__ __forceinline__ void split8(const float *v, h16x8 &hi, h16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh; lo[e] = (_Float16)((v[e] - (float)hh) * 2048.0f);
    }
}

// ... with lo itself (not scaled): the matrix pipe honours fp16 subnormals (tools/denorm_probe.py), so lo keeps an absolute
// precision of 2^-25 and all three products can share ONE accumulator (no second tile, no 2^-11 fix-up): the form the
// general kernel uses, whose register budget (nine waves per workgroup: 168) has no room for cross-term accumulators
__device__ __forceinline__ void split8u(const float *v, h16x8 &hi, h16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh; lo[e] = (_Float16)(v[e] - (float)hh);
    }
}

// SPLIT 1: `out` is the [B*L, 6*H*64] bf16 activation operand of lemon_linear_bf16x6 (the fp32 result split 3-way at the
// store); SPLIT 2: the [B*L, 3*H*64] fp16 operand of lemon_linear_f16x3
// F16: the two products as split products on the fp16 matrix cores (see k_attention_hd64_short)
template <int SPLIT, bool F16>
__global__ __launch_bounds__(576) void k_attention_hd64(const float *__restrict__ qkv, int L, int H, int causal,
                                                        float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TJ = (L + 31) >> 5;                 // key tiles
    float *sK = smem;                             // [32*TJ][PITCH]
    float *sV = smem + (size_t)32 * TJ * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.x / H;
    const int head = blockIdx.x % H;
    const int64_t tok_stride = (int64_t)3 * H * HD;            // floats between consecutive tokens
    const float *base = qkv + b * L * tok_stride + head * HD;  // q of token 0; k at +H*HD, v at +2*H*HD

    // ---- all global loads of the workgroup are issued up front: K, this lane's Q row, then V.  blockDim = 64*TJ, so every
    //      thread owns exactly 8 16-B chunks of K and 8 of V.  K goes to LDS at once; V stays in registers while the first
    //      score tile is being multiplied (its load latency hides behind those 32 MFMAs) and is parked in LDS just before
    //      the first P.V product needs it.  (zero rows beyond L: masked scores give p = 0 and 0 * 0 stays 0) ----
    float4 kreg[8], vreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + i * blockDim.x, r = id >> 4, c = id & 15;
        kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < L) kreg[i] = *reinterpret_cast<const float4 *>(base + (int64_t)r * tok_stride + 4 * c + H * HD);
    }
    // this lane's query row, columns 32h..32h+31 (the k-index pairing of MFMA #1)
    const int qi = 32 * wave + l31;
    const int qrow = qi < L ? qi : L - 1;
    float q[32];
    {
        const float *src = base + (int64_t)qrow * tok_stride + 32 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 t = *reinterpret_cast<const float4 *>(src + 4 * u);
            q[4 * u] = t.x; q[4 * u + 1] = t.y; q[4 * u + 2] = t.z; q[4 * u + 3] = t.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + i * blockDim.x, r = id >> 4, c = id & 15;
        vreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < L) vreg[i] = *reinterpret_cast<const float4 *>(base + (int64_t)r * tok_stride + 4 * c + 2 * H * HD);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + i * blockDim.x, r = id >> 4, c = id & 15;
        *reinterpret_cast<float4 *>(&sK[r * PITCH + 4 * c]) = kreg[i];
    }
    __syncthreads();

    f32x16 o0, o1;                                 // O^T tiles: output columns 0..31 and 32..63 of query qi
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    h16x8 qh[4], ql[4];
    if (F16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) split8u(q + 8 * u, qh[u], ql[u]);
    }
    float m_run = -INFINITY, l_run = 0.f;          // running max (raw dot units) and sum of this query
    const float c_exp = 0.125f * 1.44269504088896340736f;   // 1/sqrt(64) * log2(e)

    const int tj_end = causal ? (wave + 1 < TJ ? wave + 1 : TJ) : TJ;   // causal: key tiles beyond the query tile are empty
    for (int tj = 0; tj < tj_end; ++tj) {
        // S^T tile = K[32tj.., :] Q^T : lane (i, h) gets scores of keys j = 32tj + (e&3) + 8(e>>2) + 4h
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        const float *krow = &sK[(32 * tj + l31) * PITCH + 32 * h];
        if (F16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 k0 = *reinterpret_cast<const float4 *>(krow + 8 * u), k1 = *reinterpret_cast<const float4 *>(krow + 8 * u + 4);
                const float kv[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
                h16x8 kh, kl;
                split8u(kv, kh, kl);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[u], s, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 k4 = *reinterpret_cast<const float4 *>(krow + 4 * u);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, q[4 * u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, q[4 * u + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, q[4 * u + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, q[4 * u + 3], s, 0, 0, 0);
            }
        }
        // mask (padding keys, causal) and tile max
        float mt = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = 32 * tj + (e & 3) + 8 * (e >> 2) + 4 * h;
            const bool ok = j < L && (!causal || j <= qi);
            s[e] = ok ? s[e] : -INFINITY;
            mt = fmaxf(mt, s[e]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);      // finite from the first tile on (key 0 is visible to every query)
        const float alpha = exp2f((m_run - m_new) * c_exp);
        float lt = 0.f;
        // (F16: the probabilities carry a factor 2^10 -- their unscaled lo parts then keep full relative precision down to
        // p = 2^-12 -- which l_run carries as well and the final 1 / l_run removes)
        const float pbias = F16 ? 10.0f : 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = exp2f((s[e] - m_new) * c_exp + pbias);
            lt += s[e];
        }
        lt += __shfl_xor(lt, 32);
        l_run = l_run * alpha + lt;
        m_run = m_new;
        if (tj == 0) {                                // every wave passes here exactly once (tj_end >= 1): V into LDS
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int id = tid + i * blockDim.x, r = id >> 4, c = id & 15;
                *reinterpret_cast<float4 *>(&sV[r * PITCH + 4 * c]) = vreg[i];
            }
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
        // O^T += V^T P^T : k-step m pairs keys (m&3) + 8(m>>2) + 4h of the tile, i.e. s[m] as it lies
        if (F16) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {              // 16 keys per step: slot i of lane half h = key 16 t + 8 (i / 4) + 4 h + i % 4 = s[8 t + i]
                float pv[8], v0[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = s[8 * t + i];
                h16x8 ph, pl, vh, vl;
                split8u(pv, ph, pl);
#pragma unroll
                for (int i = 0; i < 8; ++i) v0[i] = sV[(32 * tj + 16 * t + 8 * (i >> 2) + 4 * h + (i & 3)) * PITCH + l31];
                split8u(v0, vh, vl);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) v0[i] = sV[(32 * tj + 16 * t + 8 * (i >> 2) + 4 * h + (i & 3)) * PITCH + 32 + l31];
                split8u(v0, vh, vl);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const float *vrow = &sV[(32 * tj + (m & 3) + 8 * (m >> 2) + 4 * h) * PITCH + l31];
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[m], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[m], o1, 0, 0, 0);
            }
        }
    }

    if (qi < L) {
        const float inv = 1.0f / l_run;
        float *dst = out + ((b * L + qi) * H + head) * HD;
        unsigned short *row6 = reinterpret_cast<unsigned short *>(out) + (b * L + qi) * lemon_split::split_segments(SPLIT == 3 ? 2 : SPLIT) * (int64_t)(H * HD);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 8 * g + 4 * h;
            const float4 v0 = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            const float4 v1 = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            if (SPLIT == 3) {          // tile-major operand of lemon_linear_f16x3t: four values = half a 16-byte slot of the row
                unsigned short *ot = reinterpret_cast<unsigned short *>(out);
                const float vv[2][4] = {{v0.x, v0.y, v0.z, v0.w}, {v1.x, v1.y, v1.z, v1.w}};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    lemon_split::us4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned short a_, b_, c_;
                        lemon_split::split2h<false>(vv[u][e], a_, b_, c_);
                        hi[e] = a_; lo[e] = c_;
                    }
                    const int64_t o = lemon_split::tiled_off(lemon_split::TILE_A_ROWS, b * L + qi, head * HD + 32 * u + c0, 0, H * HD);
                    *reinterpret_cast<lemon_split::us4 *>(ot + o) = hi;
                    *reinterpret_cast<lemon_split::us4 *>(ot + o + lemon_split::TILE_A_ROWS * 16) = lo;
                }
            } else if (SPLIT) {
                lemon_split::store_split4<(SPLIT == 1 || SPLIT == 2) ? SPLIT : 1, false>(row6, H * HD, (head * HD + c0) >> 2, v0);
                lemon_split::store_split4<(SPLIT == 1 || SPLIT == 2) ? SPLIT : 1, false>(row6, H * HD, (head * HD + 32 + c0) >> 2, v1);
            } else {
                *reinterpret_cast<float4 *>(dst + c0) = v0;
                *reinterpret_cast<float4 *>(dst + 32 + c0) = v1;
            }
        }
    }
}


// ---- long sequences (64 < L <= 288), split-fp16 arithmetic: K and V are split ONCE, at staging ---------------------------------
// k_attention_hd64<.., true> keeps K and V in LDS as fp32 and every wave re-splits the rows it reads into fp16 pairs: at
// L = 197 that is 7 x the conversion work (4 VALU instructions per element against 24 MFMAs per key tile: the kernel was
// VALU-bound, 2.06 TB/s), its Q rows are fetched one 16-byte piece per lane and line (64 lines per load instruction) and its
// output leaves as 32-byte pieces of 32 rows.  Here:
//   * the loading thread converts its K / V chunk to fp16 hi / lo planes before the LDS store (same bytes in LDS, no
//     conversion in the loop); K planes are read row-wise (ds_read_b128: 8 d of one key), V planes are read TRANSPOSED by
//     ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group, delivered key-major per d column: the V^T operand of the P.V
//     product without a transposed store and without 2-byte gathers);
//   * 128-byte rows, no padding (four planes of 288 rows = 147 KB), conflict-free by XOR: K chunk ^= (row / 2) % 8, V 64-byte
//     half ^= (row / 2) % 2 (the four rows of a transposed block then cover all 64 banks once per 32-lane half);
//   * Q goes through LDS once (fp32, in the V region before V is stored): 16 lanes fetch one token's 256 B, every lane picks up
//     its own half row; the output tile goes back through the K region and leaves 256 B (or the split operand's 128-byte
//     segments) per 16 / 8 lanes.
// Arithmetic, summation order and therefore the bits are those of k_attention_hd64<SPLIT, true> (unscaled lo parts, one
// accumulator per product, probabilities carried with a factor 2^10): tests compare the two kernels for equality.
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4v lds_fp16x4v;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4u(const float4 v, h16x4 &hi, h16x4 &lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hh = (_Float16)x[e];
        hi[e] = hh; lo[e] = (_Float16)(x[e] - (float)hh);
    }
}

// Key blocks: for 6 <= TJ <= 8 (L = 161 .. 256, e.g. the 197 tokens of ViT-B/16) the keys are staged in TWO blocks of ceil(TJ / 2)
// tiles: the four planes then take 64 KB instead of 115 KB at L = 197, so that TWO workgroups share a CU and one's staging (global
// loads, conversions, barriers) runs under the other's MFMA loop: 239 -> 214 us per 256 x 197 x 12 batch (2.9 TB/s; the first
// general kernel: 301 us).  PMC at that point (tools/r4_attn_pmc.sh): waves wait half their cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES
// 0.50), the vector ALU is busy 0.48 of the time and the matrix pipe 0.24 -- the softmax's vector work per key tile (exponentials,
// the fp16 split of the probabilities) is now what the MFMAs wait for.
template <int SPLIT>
__global__ __launch_bounds__(576, 4) void k_attention_hd64_f16(const float *__restrict__ qkv, int L, int H, int causal,
                                                            float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int TJ = (L + 31) >> 5, Lp = 32 * TJ;
    const int TB = (TJ <= 5 || TJ > 8) ? TJ : (TJ + 1) >> 1; // key tiles per block (TJ = 9: two workgroups of nine waves do not fit a CU's
                                                             // registers -- five waves on a SIMD at 128 each --, so the split would only add barriers)
    const int KB = 32 * TB;                                   // keys per block
    char *sKh = smem_c, *sKl = smem_c + KB * 128, *sVh = smem_c + 2 * KB * 128, *sVl = smem_c + 3 * KB * 128;
    float *sQ = reinterpret_cast<float *>(smem_c);           // fp32 [Lp][64] staging of Q (before the first block) and of the output
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.x / H;
    const int head = blockIdx.x % H;
    const int64_t tok_stride = (int64_t)3 * H * HD;
    const float *base = qkv + b * L * tok_stride + head * HD;
    const int nthr = 64 * TJ;                                // = blockDim.x
    // fp32 staging image: 256-byte rows, 16-byte chunk c of row r at chunk c ^ (r % 16) (conflict-free row-per-lane reads)
    auto stage_off = [](int r, int c) { return r * 64 + 4 * (c ^ (r & 15)); };
    const unsigned tok_bytes = (unsigned)(3 * H * HD * 4);     // (L <= 288 tokens of <= 2^20 bytes: offsets inside a head's slice fit 32 bits)
    constexpr int CH = 5;                                    // 16-byte chunks per thread and staging pass: one pass covers a key block
                                                             // when there are two (32 TB 16 / (64 TJ) <= 4.6), two passes the single block of TJ <= 5
    // a key block -> fp16 hi / lo planes: K and V rows [kb KB, kb KB + KB) of the head (zero beyond L), rows local to the block
    auto stage_block = [&](int kb) {
        for (int p0 = 0; p0 < KB * 16; p0 += CH * nthr) {
            // K, then V (one operand's chunks in registers at a time: with both, the second block's staging -- accumulators
            // and query fragments live -- spilled)
#pragma unroll
            for (int kv = 0; kv < 2; ++kv) {
                float4 reg[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int id = p0 + tid + i * nthr, r = kb * KB + (id >> 4), c = id & 15;
                    reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    // (wave-uniform 64-bit base + 32-bit byte offset: one address register per chunk)
                    if (id < KB * 16 && r < L)
                        reg[i] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base + (kv + 1) * H * HD) + ((unsigned)r * tok_bytes + 16u * (unsigned)c));
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int id = p0 + tid + i * nthr, r = id >> 4, c = id & 15;
                    if (id < KB * 16) {
                        h16x4 hi, lo;
                        split4u(reg[i], hi, lo);
                        const int o = kv == 0 ? r * 128 + ((((c >> 1) ^ ((r >> 1) & 7))) << 4) + (c & 1) * 8
                                              : r * 128 + ((c * 8) ^ (((r >> 1) & 1) << 6));
                        *reinterpret_cast<h16x4 *>((kv == 0 ? sKh : sVh) + o) = hi;
                        *reinterpret_cast<h16x4 *>((kv == 0 ? sKl : sVl) + o) = lo;
                    }
                }
            }
        }
    };
    // ---- Q through LDS: 16 lanes fetch one token's 256 B, every lane then picks up its own half row ----
    {
        float4 qreg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = tid + i * nthr, r = id >> 4, c = id & 15;
            qreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < L) qreg[i] = *reinterpret_cast<const float4 *>(base + (int64_t)r * tok_stride + 4 * c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = tid + i * nthr, r = id >> 4, c = id & 15;
            *reinterpret_cast<float4 *>(&sQ[stage_off(r, c)]) = qreg[i];
        }
    }
    __syncthreads();
    const int qi = 32 * wave + l31;
    h16x8 qh[4], ql[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float4 a = *reinterpret_cast<const float4 *>(&sQ[stage_off(qi, 8 * h + 2 * u)]);
        const float4 c4 = *reinterpret_cast<const float4 *>(&sQ[stage_off(qi, 8 * h + 2 * u + 1)]);
        const float qv[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
        split8u(qv, qh[u], ql[u]);
    }
    __syncthreads();                                          // every lane has its query: the buffer now takes the first key block
    stage_block(0);
    __syncthreads();

    // per-lane address parts.  K: row l31 of the key tile, chunk (4h + u) ^ ((l31 / 2) % 8).  V (transposed read, lane 4q + p of
    // the 16-lane group g supplies row q, columns 4p .. 4p+3 of the block): row 4h + q of the 8-key group, columns 16 (g % 2) + 4p
    // (+ 32 for the second output half: byte 64 = one XOR), 64-byte half ^= (row / 2) % 2 = q / 2
    int koff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) koff[u] = l31 * 128 + (((4 * h + u) ^ ((l31 >> 1) & 7)) << 4);
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int voff0 = (4 * (g16 >> 1) + q4) * 128 + (((16 * (g16 & 1) + 4 * p4) * 2) ^ ((q4 >> 1) << 6));
    const int voff1 = voff0 ^ 64;

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float c_exp = 0.125f * 1.44269504088896340736f;     // 1/sqrt(64) * log2(e)
    const int tj_end = causal ? (wave + 1 < TJ ? wave + 1 : TJ) : TJ;
    for (int kb = 0; kb * TB < TJ; ++kb) {
        if (kb > 0) {                                         // the next key block takes the buffer (every thread stages, every wave waits)
            __syncthreads();                                  // all waves are past their last read of the previous block
            stage_block(kb);
            __syncthreads();
        }
        const int t_hi = (kb + 1) * TB < tj_end ? (kb + 1) * TB : tj_end;
        for (int tj = kb * TB; tj < t_hi; ++tj) {
            const int tl = tj - kb * TB;                      // tile inside the block
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const char *kh_t = sKh + tl * 4096, *kl_t = sKl + tl * 4096;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const h16x8 kh = *reinterpret_cast<const h16x8 *>(kh_t + koff[u]);
                const h16x8 kl = *reinterpret_cast<const h16x8 *>(kl_t + koff[u]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[u], s, 0, 0, 0);
            }
            // masks only where a tile can hold a masked key: the last tile (keys >= L) and, causal, the wave's diagonal tile
            // (wave-uniform branch; the loop runs with every lane active, as the transposed reads below require)
            if (32 * tj + 32 > L || (causal && tj == wave)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = 32 * tj + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool ok = j < L && (!causal || j <= qi);
                    s[e] = ok ? s[e] : -INFINITY;
                }
            }
            float mt = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = exp2f((m_run - m_new) * c_exp);
            float lt = 0.f;
            // (v_exp_f32 directly: arguments are <= 10, a result below 2^-126 -- p < 2^-136 -- may come out as 0 instead of a
            // denormal, which neither the sums, >= 2^10, nor the fp16 parts of p can see)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = __builtin_amdgcn_exp2f((s[e] - m_new) * c_exp + 10.0f);     // probabilities x 2^10 (l_run carries the factor, 1 / l_run removes it)
                lt += s[e];
            }
            lt += __shfl_xor(lt, 32);
            l_run = l_run * alpha + lt;
            m_run = m_new;
            if (!__all(alpha == 1.0f)) {                      // the running maximum moved for some query of the wave
#pragma unroll
                for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
            }
            const char *vh_t = sVh + tl * 4096, *vl_t = sVl + tl * 4096;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = s[8 * t + i];
                h16x8 ph, pl;
                split8u(pv, ph, pl);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int vo = (half ? voff1 : voff0) + t * 2048;
                    // k-slots 0-3 = keys 16t + 4h + 0..3, k-slots 4-7 = keys 16t + 8 + 4h + 0..3 of the tile, at d = 32 half + l31
                    const fp16x4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4v *)(vh_t + vo));
                    const fp16x4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4v *)(vh_t + vo + 1024));
                    const fp16x4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4v *)(vl_t + vo));
                    const fp16x4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4v *)(vl_t + vo + 1024));
                    h16x8 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vh[e] = (_Float16)a0[e]; vh[4 + e] = (_Float16)a1[e];
                        vl[e] = (_Float16)b0[e]; vl[4 + e] = (_Float16)b1[e];
                    }
                    if (half == 0) {
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o0, 0, 0, 0);
                    } else {
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o1, 0, 0, 0);
                    }
                }
            }
        }
    }
    // the output tile goes through the buffer (every wave is past its last key tile)
    float *sO = sQ;
    __syncthreads();
    {
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 8 * g + 4 * h;                     // columns c0 .. c0+3 (and 32 + c0 ..) of query qi
            *reinterpret_cast<float4 *>(&sO[stage_off(qi, c0 >> 2)]) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(&sO[stage_off(qi, 8 + (c0 >> 2))]) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                             // 8 lanes per token row, 8 values each: 16-byte stores in every form
        const int id = tid + i * nthr;
        const int r = SPLIT == 3 ? id % Lp : id >> 3, c8 = SPLIT == 3 ? id / Lp : id & 7;
        if (r < L) {
            const float4 v0 = *reinterpret_cast<const float4 *>(&sO[stage_off(r, 2 * c8)]);
            const float4 v1 = *reinterpret_cast<const float4 *>(&sO[stage_off(r, 2 * c8 + 1)]);
            if (SPLIT == 3)
                lemon_split::store_tiled8<lemon_split::TILE_A_ROWS, false>(reinterpret_cast<unsigned short *>(out), b * L + r, H * HD, head * (HD / 8) + c8, v0, v1);
            else if (SPLIT)
                lemon_split::store_split8<(SPLIT == 1 || SPLIT == 2) ? SPLIT : 1, false>(reinterpret_cast<unsigned short *>(out) + (b * L + r) * lemon_split::split_segments(SPLIT == 3 ? 2 : SPLIT) * (int64_t)(H * HD), H * HD,
                                                 head * (HD / 8) + c8, v0, v1);
            else {
                float *dst = out + ((b * L + r) * H + head) * HD + 8 * c8;
                *reinterpret_cast<float4 *>(dst) = v0;
                *reinterpret_cast<float4 *>(dst + 4) = v1;
            }
        }
    }
}


// Short sequences (L <= 64: the 50 tokens of a ViT-B/32 image, the 8..64 tokens of a prompt batch): ONE LDS buffer of
// 32*TJ rows serves K first and V afterwards -- all score tiles of a query fit in registers, so the softmax is computed
// on the complete row (no running rescale) and K is dead by the time V is needed.  Half the LDS of the general kernel
// (17 KB at TJ = 2): the CU holds six workgroups instead of four, which is what this latency-bound shape was short of
// (MFMA pipe busy 0.40, 3.6 TB/s with four).
// F16: both products on the fp16 matrix cores as split products (hi.hi in one accumulator, lo.hi + hi.lo in a second one that
// enters with 2^-11; the dropped lo.lo is 2^-22 of a product): the fp32 form is BOUND by v_mfma_f32_32x32x2_f32 -- 256 of them,
// 64 cycles each, per wave = 437 us of matrix-pipe time per 131 000-token micro-batch, exactly what the kernel took --, the split
// form needs 48 MFMAs of 32 cycles and leaves the kernel to its memory traffic.
template <int TJ, int SPLIT, bool F16>
__global__ __launch_bounds__(64 * TJ, 3) void k_attention_hd64_short(const float *__restrict__ qkv, int L, int H, int causal,
                                                                  float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sKV[32 * TJ * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.x / H;
    const int head = blockIdx.x % H;
    const int64_t tok_stride = (int64_t)3 * H * HD;
    const float *base = qkv + b * L * tok_stride + head * HD;
    float4 kreg[8], vreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + i * 64 * TJ, r = id >> 4, c = id & 15;
        kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < L) kreg[i] = *reinterpret_cast<const float4 *>(base + (int64_t)r * tok_stride + 4 * c + H * HD);
    }
    const int qi = 32 * wave + l31;
    // Q like K and V: 16 lanes fetch one token's 256 B (a lane walking its OWN row 16 B at a time touches 64 different
    // lines per load instruction), then the tile goes through the LDS buffer once so that every lane can pick up its
    // query's half row.  Rows >= L are zero (their queries are never stored).
    {
        float4 qreg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = tid + i * 64 * TJ, r = id >> 4, c = id & 15;
            qreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < L) qreg[i] = *reinterpret_cast<const float4 *>(base + (int64_t)r * tok_stride + 4 * c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = tid + i * 64 * TJ, r = id >> 4, c = id & 15;
            *reinterpret_cast<float4 *>(&sKV[r * PITCH + 4 * c]) = qreg[i];
        }
    }
    __syncthreads();
    float q[32];
    {
        const float *src = &sKV[qi * PITCH + 32 * h];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 t = *reinterpret_cast<const float4 *>(src + 4 * u);
            q[4 * u] = t.x; q[4 * u + 1] = t.y; q[4 * u + 2] = t.z; q[4 * u + 3] = t.w;
        }
    }
    __syncthreads();                                // every lane has its query row: the buffer now takes K
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + i * 64 * TJ, r = id >> 4, c = id & 15;
        *reinterpret_cast<float4 *>(&sKV[r * PITCH + 4 * c]) = kreg[i];
    }
    __syncthreads();

    // V is fetched only now, under the score MFMAs: holding it in registers from the start (184 VGPRs) kept a SIMD at two
    // waves; its latency is covered by the other workgroups of the CU instead
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + i * 64 * TJ, r = id >> 4, c = id & 15;
        vreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < L) vreg[i] = *reinterpret_cast<const float4 *>(base + (int64_t)r * tok_stride + 4 * c + 2 * H * HD);
    }
    // all score tiles of this lane's query: S^T tile tj = K[32tj.., :] Q^T (two accumulators per tile: even / odd k-steps,
    // so that consecutive MFMAs do not wait for each other)
    f32x16 s[TJ];
    h16x8 qh[4], ql[4];                              // F16: the query's half row as four 8-k fragments, hi and lo parts
    if (F16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) split8(q + 8 * u, qh[u], ql[u]);
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        f32x16 sa, sb;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
        const float *krow = &sKV[(32 * tj + l31) * PITCH + 32 * h];
        if (F16) {
            // k-slot i of lane half h in step u is d = 32 h + 8 u + i for both operands; sa = hi.hi, sb = (lo.hi + hi.lo) 2^11
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 k0 = *reinterpret_cast<const float4 *>(krow + 8 * u), k1 = *reinterpret_cast<const float4 *>(krow + 8 * u + 4);
                const float kv[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
                h16x8 kh, kl;
                split8(kv, kh, kl);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[u], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[u], sb, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[u], sb, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) s[tj][e] = sa[e] + sb[e] * 0.00048828125f;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 k4 = *reinterpret_cast<const float4 *>(krow + 4 * u);
                sa = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, q[4 * u], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, q[4 * u + 1], sb, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, q[4 * u + 2], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, q[4 * u + 3], sb, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) s[tj][e] = sa[e] + sb[e];
        }
    }
    // mask (padding keys, causal), row maximum, exponentials, row sum
    const float c_exp = 0.125f * 1.44269504088896340736f;   // 1/sqrt(64) * log2(e)
    float mx = -INFINITY;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = 32 * tj + (e & 3) + 8 * (e >> 2) + 4 * h;
            const bool ok = j < L && (!causal || j <= qi);
            s[tj][e] = ok ? s[tj][e] : -INFINITY;
            mx = fmaxf(mx, s[tj][e]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));           // finite: key 0 is visible to every query
    float lsum = 0.f;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[tj][e] = exp2f((s[tj][e] - mx) * c_exp);
            lsum += s[tj][e];
        }
    lsum += __shfl_xor(lsum, 32);

    __syncthreads();                                // every wave is done with K
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + i * 64 * TJ, r = id >> 4, c = id & 15;
        *reinterpret_cast<float4 *>(&sKV[r * PITCH + 4 * c]) = vreg[i];
    }
    __syncthreads();

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    if (F16) {
        // O^T = V^T P^T, 16 keys per step: k-slot i of lane half h in step t of key tile tj is key 32 tj + 16 t + 8 (i / 4) + 4 h
        // + i % 4 -- exactly the accumulator elements 8 t .. 8 t + 7 the lane holds of P, and the same slots for V's fragments
        f32x16 c0, c1;                                // the 2^11-scaled cross terms
#pragma unroll
        for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pv[8], v0[8], v1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pv[i] = s[tj][8 * t + i];
                    const float *vrow = &sKV[(32 * tj + 16 * t + 8 * (i >> 2) + 4 * h + (i & 3)) * PITCH + l31];
                    v0[i] = vrow[0]; v1[i] = vrow[32];
                }
                h16x8 ph, pl, vh, vl;
                split8(pv, ph, pl);
                split8(v0, vh, vl);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, c0, 0, 0, 0);
                split8(v1, vh, vl);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, c1, 0, 0, 0);
            }
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] += c0[e] * 0.00048828125f; o1[e] += c1[e] * 0.00048828125f; }
    } else {
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const float *vrow = &sKV[(32 * tj + (m & 3) + 8 * (m >> 2) + 4 * h) * PITCH + l31];
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[tj][m], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[tj][m], o1, 0, 0, 0);
            }
    }
    // the output tile goes back through the LDS buffer (V is dead): 16 lanes then store one token's 256 B (or, SPLIT, the
    // six 128-B bf16 segments of it) instead of 32-B pieces of 32 different rows per store instruction
    __syncthreads();
    {
        const float inv = 1.0f / lsum;
        float *dst = &sKV[qi * PITCH];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 8 * g + 4 * h;
            *reinterpret_cast<float4 *>(dst + c0) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(dst + 32 + c0) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                   // 8 lanes per token row, 8 values each: 16-byte stores in both forms
        const int id = tid + i * 64 * TJ;
        // (tile-major operand: the 16-byte slots of consecutive ROWS are adjacent, so consecutive lanes take consecutive rows)
        const int r = SPLIT == 3 ? id % (32 * TJ) : id >> 3, c8 = SPLIT == 3 ? id / (32 * TJ) : id & 7;
        if (r < L) {
            const float4 v0 = *reinterpret_cast<const float4 *>(&sKV[r * PITCH + 8 * c8]);
            const float4 v1 = *reinterpret_cast<const float4 *>(&sKV[r * PITCH + 8 * c8 + 4]);
            if (SPLIT == 3)
                lemon_split::store_tiled8<lemon_split::TILE_A_ROWS, false>(reinterpret_cast<unsigned short *>(out), b * L + r, H * HD, head * (HD / 8) + c8, v0, v1);
            else if (SPLIT)
                lemon_split::store_split8<(SPLIT == 1 || SPLIT == 2) ? SPLIT : 1, false>(reinterpret_cast<unsigned short *>(out) + (b * L + r) * lemon_split::split_segments(SPLIT == 3 ? 2 : SPLIT) * (int64_t)(H * HD), H * HD,
                                                 head * (HD / 8) + c8, v0, v1);
            else {
                float *dst = out + ((b * L + r) * H + head) * HD + 8 * c8;
                *reinterpret_cast<float4 *>(dst) = v0;
                *reinterpret_cast<float4 *>(dst + 4) = v1;
            }
        }
    }
}

}  // namespace

// Arithmetic of the two products: 1 = split products on the fp16 matrix cores (q, k, v and the probabilities carried as fp16
// pairs: inputs must stay inside +-65 504, see include/lemon_hip.h), 0 = v_mfma_f32_32x32x2_f32 (no range limit).
// Per calling THREAD (round 5: it was one unsynchronised process-wide int, so two embedders in different GEMM modes -- or a bf16x6
// re-embedding next to an f16x3 pass -- could pick each other's kernel): set by lemon_attention_set_f16, which the host mirror
// calls in front of every tower pass; $LEMON_ATTN_F16=0 starts every thread with 0.
static int attn_f16_default() {
    static const int v = [] { const char *e = getenv("LEMON_ATTN_F16"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}
static thread_local int g_attn_f16 = attn_f16_default();

static thread_local int g_attn_old_general = 0;   // lemon_attention_set_f16(2): fp16 arithmetic with the FIRST general kernel (tests: bit equality)

extern "C" int lemon_attention_set_f16(int on) {
    g_attn_old_general = on == 2;
    const int prev = g_attn_f16;
    g_attn_f16 = on ? 1 : 0;
    return prev;
}

template <int SPLIT>
static int attention_impl(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                          int causal, float *out_dev, void *stream) {
    LEMON_REQUIRE(batch >= 0 && seq_len > 0 && heads > 0, "batch >= 0, seq_len > 0, heads > 0");
    LEMON_REQUIRE(head_dim == HD, "head_dim must be 64");
    LEMON_REQUIRE(seq_len <= 288, "seq_len <= 288 (K and V of one head are staged in LDS)");
    if (batch == 0) return LEMON_OK;
    LEMON_REQUIRE(qkv_dev && out_dev, "null pointer");
    LEMON_REQUIRE((((uintptr_t)qkv_dev) & 15) == 0 && (((uintptr_t)out_dev) & 15) == 0, "16-byte alignment");
    LEMON_REQUIRE(batch * heads < (int64_t)1 << 31, "batch * heads < 2^31");
    const int tj = (seq_len + 31) / 32;
    static const bool short_off = [] { const char *e = getenv("LEMON_ATTN_SHORT"); return e && e[0] == '0'; }();   // tuning knob
    const dim3 grid((unsigned)(batch * heads));
    if (tj <= 2 && !short_off) {
        const bool f16_off = g_attn_f16 == 0;
        if (f16_off) {
            if (tj == 1) hipLaunchKernelGGL((k_attention_hd64_short<1, SPLIT, false>), grid, dim3(64), 0, (hipStream_t)stream, qkv_dev, seq_len, heads, causal, out_dev);
            else         hipLaunchKernelGGL((k_attention_hd64_short<2, SPLIT, false>), grid, dim3(128), 0, (hipStream_t)stream, qkv_dev, seq_len, heads, causal, out_dev);
        } else {
            if (tj == 1) hipLaunchKernelGGL((k_attention_hd64_short<1, SPLIT, true>), grid, dim3(64), 0, (hipStream_t)stream, qkv_dev, seq_len, heads, causal, out_dev);
            else         hipLaunchKernelGGL((k_attention_hd64_short<2, SPLIT, true>), grid, dim3(128), 0, (hipStream_t)stream, qkv_dev, seq_len, heads, causal, out_dev);
        }
        LEMON_HIP_CHECK(hipGetLastError());
        return LEMON_OK;
    }
    // 64 < L <= 288.  Split-fp16 arithmetic: the kernel that splits K and V once at staging (k_attention_hd64_f16);
    // LEMON_ATTN_GENERAL=old keeps the first version (fp32 K / V in LDS, re-split by every wave) for A/B runs and the equality test
    static const bool old_general = [] { const char *e = getenv("LEMON_ATTN_GENERAL"); return e && !strcmp(e, "old"); }();
    const bool staged = g_attn_f16 != 0 && !old_general && !g_attn_old_general;
    // staged kernel: four planes of one key block (ceil(tj / 2) tiles when tj > 5), but never less than the fp32 [32 tj][64]
    // image Q and the output pass through
    const int tb = (tj <= 5 || tj > 8) ? tj : (tj + 1) / 2;
    const size_t lds_staged = std::max((size_t)4 * 32 * tb * 128, (size_t)32 * tj * 256);
    const size_t lds = staged ? lds_staged : (size_t)2 * 32 * tj * PITCH * sizeof(float);
    {   // the attribute is per DEVICE (and per instantiation): one flag per device index, under a lock
        static std::mutex mu;
        static bool attr_set[64] = {};
        int dev = 0;
        LEMON_HIP_CHECK(hipGetDevice(&dev));
        LEMON_REQUIRE(dev >= 0 && dev < 64, "device index");
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_attention_hd64<SPLIT, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_attention_hd64<SPLIT, false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_attention_hd64_f16<SPLIT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    if (staged) {
        hipLaunchKernelGGL((k_attention_hd64_f16<SPLIT>), grid, dim3(64 * tj), lds, (hipStream_t)stream, qkv_dev, seq_len, heads, causal, out_dev);
        LEMON_HIP_CHECK(hipGetLastError());
        return LEMON_OK;
    }
    const bool f16_off_g = g_attn_f16 == 0;
    if (f16_off_g) hipLaunchKernelGGL((k_attention_hd64<SPLIT, false>), grid, dim3(64 * tj), lds, (hipStream_t)stream, qkv_dev, seq_len, heads, causal, out_dev);
    else hipLaunchKernelGGL((k_attention_hd64<SPLIT, true>), grid, dim3(64 * tj), lds, (hipStream_t)stream, qkv_dev, seq_len, heads, causal, out_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_attention_f32(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                                   int causal, float *out_dev, void *stream) {
    return attention_impl<0>(qkv_dev, batch, seq_len, heads, head_dim, causal, out_dev, stream);
}

// lemon_attention_f32 whose output is the split activation operand of lemon_linear_bf16x6: out6_dev [batch*seq_len, 6*heads*64] bf16
extern "C" int lemon_attention_split3(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                                      int causal, uint16_t *out6_dev, void *stream) {
    return attention_impl<1>(qkv_dev, batch, seq_len, heads, head_dim, causal, reinterpret_cast<float *>(out6_dev), stream);
}

// ... of lemon_linear_f16x3: out3_dev [batch*seq_len, 3*heads*64] fp16
extern "C" int lemon_attention_f16x3(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                                     int causal, uint16_t *out3_dev, void *stream) {
    return attention_impl<2>(qkv_dev, batch, seq_len, heads, head_dim, causal, reinterpret_cast<float *>(out3_dev), stream);
}

// ... as the tile-major fp16 activation operand of lemon_linear_f16x3t (the output projection in the hand-written GEMM):
// outt_dev holds ceil(batch*seq_len / 128) * 128 x heads*64 x 2 halves, 16-byte aligned
extern "C" int lemon_attention_f16x3t(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                                      int causal, uint16_t *outt_dev, void *stream) {
    return attention_impl<3>(qkv_dev, batch, seq_len, heads, head_dim, causal, reinterpret_cast<float *>(outt_dev), stream);
}
