// split3.hpp -- device helpers of the split GEMM operands (lemon_linear_bf16x6: 3-way bf16, six products; lemon_linear_f16x3:
// 2-way fp16, three products), shared by encoder.hip and attention.hip.  SCHEME 1 = bf16x6, 2 = f16x3 throughout.
#pragma once
#include <hip/hip_runtime.h>

namespace lemon_split {

// ---- 3-way bf16 split of fp32 values (lemon_linear_bf16x6) ---------------------------------------------------------
// v = hi + mid + lo with hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): both differences are exact in fp32, so
// the three parts carry 24 significant bits of v (|v - hi - mid - lo| <= 2^-25 |v|).  A row of the ACTIVATION operand is
// stored as six k-long bf16 segments [hi | hi | mid | hi | mid | lo], a row of the WEIGHT operand as
// [hi | mid | hi | lo | mid | hi]: the dot product of the two rows is the sum of the six cross products of order <= 2.
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float bf16_val(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ void split3(float v, unsigned short &hi, unsigned short &mid, unsigned short &lo) {
    // (no FMA contraction: when the caller computed v as a product, `v - hi` would otherwise become fma(a, b, -hi) and the
    // parts would describe the UNROUNDED product -- a producer with a fused split and the same producer followed by
    // lemon_split3_f32 must store the same bits)
#pragma clang fp contract(off)
    asm volatile("" : "+v"(v));
    hi = bf16_bits(v);
    const float r1 = v - bf16_val(hi);
    mid = bf16_bits(r1);
    lo = bf16_bits(r1 - bf16_val(mid));
}
// ---- 2-way fp16 split (lemon_linear_f16x3) -------------------------------------------------------------------------
// v = hi + lo with hi = f16(v) and lo = v - hi (exact in fp32, |lo| <= 2^-11 |v|), kept as f16(lo * 2^11): 11 + 11 significant
// bits and the sign of lo -- |v - hi - lo| <= 2^-23 |v|, exact for half of all fp32 values.  An ACTIVATION row is stored as
// three k-long fp16 segments [hi | hi | lo*2^11], a WEIGHT row as [hi | lo | hi*2^-11] of w * wscale (wscale a power of two
// that lifts the tensor's largest magnitude to 2^14..2^15, so that lo and hi*2^-11 of every weight that matters are fp16
// normals; the GEMM's alpha carries 1 / wscale).  The row dot product is hi.hi + hi.lo + lo.hi; the dropped lo.lo term is
// <= 2^-22 of the product (2^-26 typically).  Values beyond the fp16 range (|v| >= 65 520) become inf and the GEMM result
// NaN: loud, not wrong.
__device__ __forceinline__ unsigned short f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ float f16_val(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
template <bool WEIGHT>
__device__ __forceinline__ void split2h(float v, unsigned short &a, unsigned short &b, unsigned short &c) {
#pragma clang fp contract(off)
    asm volatile("" : "+v"(v));
    const unsigned short hi = f16_bits(v);
    const float lo = v - f16_val(hi);
    if (WEIGHT) { a = hi; b = f16_bits(lo); c = f16_bits(f16_val(hi) * 0.00048828125f); }
    else        { a = hi; b = hi; c = f16_bits(lo * 2048.0f); }
}
__device__ __host__ constexpr int split_segments(int scheme) { return scheme == 2 ? 3 : 6; }

// four consecutive k of one row -> the row's segments (row = start of the row's 6k bf16 / 3k fp16, c = float4 chunk index)
template <int SCHEME, bool WEIGHT> __device__ __forceinline__ void store_split4(unsigned short *__restrict__ row, int k, int c, float4 v, float wscale = 1.0f);
template <int SCHEME, bool WEIGHT> __device__ __forceinline__ void store_split8(unsigned short *__restrict__ row, int k, int c8, float4 v0, float4 v1, float wscale = 1.0f);

typedef unsigned short us8 __attribute__((ext_vector_type(8)));
template <bool WEIGHT>
__device__ __forceinline__ void store_split4_h(unsigned short *__restrict__ row3, int k, int c, float4 v, float wscale) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    us4 a, b, d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned short a_, b_, c_;
        split2h<WEIGHT>(WEIGHT ? x[e] * wscale : x[e], a_, b_, c_);
        a[e] = a_; b[e] = b_; d[e] = c_;
    }
    us4 *o = reinterpret_cast<us4 *>(row3) + c;
    const int seg = k >> 2;
    o[0] = a; o[seg] = b; o[2 * seg] = d;
}
template <bool WEIGHT>
__device__ __forceinline__ void store_split8_h(unsigned short *__restrict__ row3, int k, int c8, float4 v0, float4 v1, float wscale) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    us8 a, b, d;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        unsigned short a_, b_, c_;
        split2h<WEIGHT>(WEIGHT ? x[e] * wscale : x[e], a_, b_, c_);
        a[e] = a_; b[e] = b_; d[e] = c_;
    }
    us8 *o = reinterpret_cast<us8 *>(row3) + c8;
    const int seg = k >> 3;
    o[0] = a; o[seg] = b; o[2 * seg] = d;
}

template <bool WEIGHT>
__device__ __forceinline__ void store_split4_b(unsigned short *__restrict__ row6, int k, int c, float4 v) {
    unsigned short h_[4], m_[4], l_[4];
    split3(v.x, h_[0], m_[0], l_[0]); split3(v.y, h_[1], m_[1], l_[1]);
    split3(v.z, h_[2], m_[2], l_[2]); split3(v.w, h_[3], m_[3], l_[3]);
    const us4 hi = {h_[0], h_[1], h_[2], h_[3]}, mid = {m_[0], m_[1], m_[2], m_[3]}, lo = {l_[0], l_[1], l_[2], l_[3]};
    us4 *o = reinterpret_cast<us4 *>(row6) + c;
    const int seg = k >> 2;                               // us4 chunks per segment
    if (WEIGHT) { o[0] = hi; o[seg] = mid; o[2 * seg] = hi; o[3 * seg] = lo; o[4 * seg] = mid; o[5 * seg] = hi; }
    else        { o[0] = hi; o[seg] = hi; o[2 * seg] = mid; o[3 * seg] = hi; o[4 * seg] = mid; o[5 * seg] = lo; }
}


// eight consecutive k of one row -> the six segments, 16-byte stores (c8 = 8-element chunk index)
template <bool WEIGHT>
__device__ __forceinline__ void store_split8_b(unsigned short *__restrict__ row6, int k, int c8, float4 v0, float4 v1) {
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    us8 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        unsigned short h_, m_, l_;
        split3(v[e], h_, m_, l_);
        hi[e] = h_; mid[e] = m_; lo[e] = l_;
    }
    us8 *o = reinterpret_cast<us8 *>(row6) + c8;
    const int seg = k >> 3;                               // us8 chunks per segment
    if (WEIGHT) { o[0] = hi; o[seg] = mid; o[2 * seg] = hi; o[3 * seg] = lo; o[4 * seg] = mid; o[5 * seg] = hi; }
    else        { o[0] = hi; o[seg] = hi; o[2 * seg] = mid; o[3 * seg] = hi; o[4 * seg] = mid; o[5 * seg] = lo; }
}

template <int SCHEME, bool WEIGHT>
__device__ __forceinline__ void store_split4(unsigned short *__restrict__ row, int k, int c, float4 v, float wscale) {
    if (SCHEME == 2) store_split4_h<WEIGHT>(row, k, c, v, wscale);
    else store_split4_b<WEIGHT>(row, k, c, v);
}
template <int SCHEME, bool WEIGHT>
__device__ __forceinline__ void store_split8(unsigned short *__restrict__ row, int k, int c8, float4 v0, float4 v1, float wscale) {
    if (SCHEME == 2) store_split8_h<WEIGHT>(row, k, c8, v0, v1, wscale);
    else store_split8_b<WEIGHT>(row, k, c8, v0, v1);
}

// ---- tile-major fp16 split operands of the hand-written GEMM (gemm_f16x3.hip, lemon_linear_f16x3t) ------------------------
// Activations: [row tile of 128][k16 step][segment: hi, lo 2^11][row block of 32][k half][row in block][8 k] -- every 32-row x
// 16-k block is 1 KB in MFMA fragment order (lane = k half * 32 + row), a stage of the GEMM's ring is one contiguous copy.
// Weights: the same with 256-row tiles and segments hi, lo of w * wscale (hi 2^-11 is made in registers by the GEMM).
constexpr int TILE_A_ROWS = 128, TILE_W_ROWS = 256;
__device__ __host__ inline int64_t tiled_off(int tile_rows, int64_t row, int k, int seg, int width) {      // in halves
    const int64_t tile = row / tile_rows;
    const int r = (int)(row - tile * tile_rows);
    return ((tile * (width >> 4) + (k >> 4)) * 2 + seg) * (int64_t)(tile_rows * 16) + (r >> 5) * 512 + ((k >> 3) & 1) * 256 + (r & 31) * 8 + (k & 7);
}
// eight consecutive k (chunk c8) of one row -> the two 16-byte slots of its hi and lo parts
template <int TILE_ROWS, bool WEIGHT>
__device__ __forceinline__ void store_tiled8(unsigned short *__restrict__ base, int64_t row, int width, int c8, float4 v0, float4 v1, float wscale = 1.0f) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    us8 a, d;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        unsigned short a_, b_, c_;
        split2h<WEIGHT>(WEIGHT ? x[e] * wscale : x[e], a_, b_, c_);
        a[e] = a_; d[e] = WEIGHT ? b_ : c_;          // weights: lo itself; activations: lo * 2^11
    }
    const int64_t o = tiled_off(TILE_ROWS, row, 8 * c8, 0, width);
    *reinterpret_cast<us8 *>(base + o) = a;
    *reinterpret_cast<us8 *>(base + o + TILE_ROWS * 16) = d;
}

}  // namespace lemon_split
