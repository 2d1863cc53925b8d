// split3.hpp -- device helpers of the 3-way bf16 split operands (lemon_linear_bf16x6), shared by encoder.hip and attention.hip
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
// four consecutive k of one row -> the six segments (row6 = start of the row's 6k bf16, c = float4 chunk index)
template <bool WEIGHT>
__device__ __forceinline__ void store_split4(unsigned short *__restrict__ row6, int k, int c, float4 v) {
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
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
template <bool WEIGHT>
__device__ __forceinline__ void store_split8(unsigned short *__restrict__ row6, int k, int c8, float4 v0, float4 v1) {
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

}  // namespace lemon_split
