// encoder.hip -- row-wise pieces of the CLIP towers that sit between the library GEMMs (gfx950 only).
//
// LayerNorm of the pre-LN transformer blocks (HF CLIPEncoderLayer.layer_norm1/2, pre_layrnorm, post_layernorm,
// final_layer_norm; lib/models/chexzero_clip.py:177-183,207-212): y = (x - mean) / sqrt(var + eps) * w + b over the last
// dimension, float32.  HBM-bound: 8 bytes per element.  One wavefront per row, the row held in registers (16-B loads,
// lane l owns chunks l, l+64, ...), mean and the centred second moment reduced across the wave with two butterfly
// passes -- one read, one write, no LDS, no second pass over memory.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "split3.hpp"

using namespace lemon_split;

namespace {

template <int SCHEME, bool WEIGHT>
__global__ __launch_bounds__(256) void k_split3_rows(const float *__restrict__ x, int64_t rows, int k, unsigned short *__restrict__ y6, float wscale) {
    const int nch = k >> 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * nch) return;
    const int64_t r = t / nch;
    const int c = (int)(t - r * nch);
    store_split4<SCHEME, WEIGHT>(y6 + r * split_segments(SCHEME) * (int64_t)k, k, c, reinterpret_cast<const float4 *>(x + r * (int64_t)k)[c], wscale);
}

template <int CH, int SPLIT = 0>   // float4 chunks per lane: width <= 256*CH; SPLIT 1 / 2: y is the [rows, 6 width] bf16 / [rows, 3 width] fp16 activation operand
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ b, float eps, int64_t rows, int width,
                                                   float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = width >> 2;
    const float4 *xr = reinterpret_cast<const float4 *>(x + row * (int64_t)width);
    float4 v[CH];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nch ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)width;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (lane + 64 * i < nch) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)width + eps);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *yr = reinterpret_cast<float4 *>(y + row * (int64_t)width);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 ww = w4[c], bb = b4[c];
            float4 o;
            o.x = (v[i].x - mean) * rstd * ww.x + bb.x;
            o.y = (v[i].y - mean) * rstd * ww.y + bb.y;
            o.z = (v[i].z - mean) * rstd * ww.z + bb.z;
            o.w = (v[i].w - mean) * rstd * ww.w + bb.w;
            if (SPLIT) store_split4<SPLIT ? SPLIT : 1, false>(reinterpret_cast<unsigned short *>(y) + row * split_segments(SPLIT) * (int64_t)width, width, c, o);
            else yr[c] = o;
        }
    }
}

// LayerNorm for widths that are multiples of 8: every lane owns EIGHT consecutive elements per chunk (two float4), so the
// split variant stores 16 bytes per segment instead of 8 (94 -> see DESIGN us per 50 000 x 768 rows); the fp32 variant uses
// the same element-to-lane mapping, so lemon_layernorm_split3 == lemon_split3_f32(lemon_layernorm_f32) bit for bit.
template <int CH8, int SPLIT>   // 8-element chunks per lane: width <= 512*CH8
__global__ __launch_bounds__(256) void k_layernorm8(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ b, float eps, int64_t rows, int width,
                                                    float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = width >> 3;
    const float4 *xr = reinterpret_cast<const float4 *>(x + row * (int64_t)width);
    float4 v[CH8][2];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        const int c = lane + 64 * i;
        v[i][0] = v[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) { v[i][0] = xr[2 * c]; v[i][1] = xr[2 * c + 1]; }
        s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)width;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float dx = v[i][hf].x - mean, dy = v[i][hf].y - mean, dz = v[i][hf].z - mean, dw = v[i][hf].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)width + eps);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *yr = reinterpret_cast<float4 *>(y + row * (int64_t)width);
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float4 o[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float4 ww = w4[2 * c + hf], bb = b4[2 * c + hf];
                o[hf].x = (v[i][hf].x - mean) * rstd * ww.x + bb.x;
                o[hf].y = (v[i][hf].y - mean) * rstd * ww.y + bb.y;
                o[hf].z = (v[i][hf].z - mean) * rstd * ww.z + bb.z;
                o[hf].w = (v[i][hf].w - mean) * rstd * ww.w + bb.w;
            }
            if (SPLIT) store_split8<SPLIT ? SPLIT : 1, false>(reinterpret_cast<unsigned short *>(y) + row * split_segments(SPLIT) * (int64_t)width, width, c, o[0], o[1]);
            else { yr[2 * c] = o[0]; yr[2 * c + 1] = o[1]; }
        }
    }
}

// LayerNorm -> tile-major fp16 operand of lemon_linear_f16x3t (split3.hpp: tiled_off).  In that layout the 16-byte slots of
// CONSECUTIVE ROWS are adjacent, so the wave-per-row mapping above would scatter 16-byte pieces (measured: 125 us against
// 57 us for the row-major operand at 50 000 x 768).  Here a workgroup normalises eight consecutive rows (two per wave), parks
// their hi / lo chunks in LDS as [chunk][part][row] and writes them out eight rows = one 128-byte line per chunk and part.
// Same arithmetic and element-to-lane mapping as k_layernorm8: the operand holds exactly the split of lemon_layernorm_f32.
// RAW: the operand holds the split of x itself and aff[row] = (rstd, -mean rstd): the input side of a LayerNorm folded into the
// GEMM (lemon_linear_f16x3t_ln) for the first block of a tower, whose input no GEMM epilogue produced.
template <int CH8, bool RAW = false>
__global__ __launch_bounds__(256) void k_layernorm8_t(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ b, float eps, int64_t rows, int width,
                                                      unsigned short *__restrict__ yt, float2 *__restrict__ aff = nullptr) {
    extern __shared__ __attribute__((aligned(16))) us8 s_t[];          // [8 rows][2 parts x width/8 chunks (+1: odd pitch)] 16-byte slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = width >> 3;
    const int64_t row0 = (int64_t)blockIdx.x * 8;
    for (int rr = 0; rr < 2; ++rr) {
        const int rw = 2 * wave + rr;
        const int64_t row = row0 + rw;
        if (row >= rows) continue;                                     // (wave-uniform)
        const float4 *xr = reinterpret_cast<const float4 *>(x + row * (int64_t)width);
        float4 v[CH8][2];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            const int c = lane + 64 * i;
            v[i][0] = v[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nch) { v[i][0] = xr[2 * c]; v[i][1] = xr[2 * c + 1]; }
            s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s / (float)width;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float dx = v[i][hf].x - mean, dy = v[i][hf].y - mean, dz = v[i][hf].z - mean, dw = v[i][hf].w - mean;
                    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
        const float rstd = rsqrtf(q / (float)width + eps);
        const float4 *w4 = reinterpret_cast<const float4 *>(w);
        const float4 *b4 = reinterpret_cast<const float4 *>(b);
        if (RAW && lane == 0) {
            const bool far = fabsf(mean) * rstd > LEMON_LN_FOLD_MAX_SHIFT;   // (see common.hpp: the caller falls back)
            aff[row] = far ? make_float2(__builtin_nanf(""), __builtin_nanf("")) : make_float2(rstd, -mean * rstd);
        }
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float o[8];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if (RAW) {
                        o[4 * hf] = v[i][hf].x; o[4 * hf + 1] = v[i][hf].y; o[4 * hf + 2] = v[i][hf].z; o[4 * hf + 3] = v[i][hf].w;
                    } else {
                        const float4 ww = w4[2 * c + hf], bb = b4[2 * c + hf];
                        o[4 * hf] = (v[i][hf].x - mean) * rstd * ww.x + bb.x;
                        o[4 * hf + 1] = (v[i][hf].y - mean) * rstd * ww.y + bb.y;
                        o[4 * hf + 2] = (v[i][hf].z - mean) * rstd * ww.z + bb.z;
                        o[4 * hf + 3] = (v[i][hf].w - mean) * rstd * ww.w + bb.w;
                    }
                }
                us8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    unsigned short a_, b_, c_;
                    split2h<false>(o[e], a_, b_, c_);
                    hi[e] = a_; lo[e] = c_;
                }
                s_t[rw * (2 * nch + 1) + c] = hi;                       // (consecutive lanes, consecutive slots)
                s_t[rw * (2 * nch + 1) + nch + c] = lo;
            }
        }
    }
    __syncthreads();
    const int nrow = rows - row0 < 8 ? (int)(rows - row0) : 8;
    for (int idx = threadIdx.x; idx < nch * 16; idx += 256) {
        const int rw = idx & 7, part = (idx >> 3) & 1, c = idx >> 4;
        if (rw < nrow)
            *reinterpret_cast<us8 *>(yt + tiled_off(TILE_A_ROWS, row0 + rw, 8 * c, part, width)) = s_t[rw * (2 * nch + 1) + part * nch + c];
    }
}

// Token assembly of the vision tower (HF CLIPVisionEmbeddings + pre_layrnorm; chexzero_clip.py:243-249): row 0 of every image
// is the class embedding, rows 1.. are the patch-embedding GEMM's output; add the position embedding and apply the
// pre-LayerNorm -- torch runs this as cat + add + layer_norm (three read+write passes); here it is one.
template <int CH>
__global__ __launch_bounds__(256) void k_vision_tokens_ln(const float *__restrict__ patches, const float *__restrict__ cls,
                                                          const float *__restrict__ pos, const float *__restrict__ w,
                                                          const float *__restrict__ b, float eps, int64_t batch, int n_tok,
                                                          int width, float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= batch * n_tok) return;
    const int64_t img = row / n_tok;
    const int t = (int)(row - img * n_tok);
    const int nch = width >> 2;
    const float4 *src = t == 0 ? reinterpret_cast<const float4 *>(cls)
                               : reinterpret_cast<const float4 *>(patches + (img * (n_tok - 1) + (t - 1)) * (int64_t)width);
    const float4 *pr = reinterpret_cast<const float4 *>(pos + (int64_t)t * width);
    float4 v[CH];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 a = src[c], p = pr[c];
            v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (w == nullptr) {                                   // (kernel-uniform) token assembly only: no LayerNorm in front of the blocks
        float4 *yo = reinterpret_cast<float4 *>(y + row * (int64_t)width);
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (lane + 64 * i < nch) yo[lane + 64 * i] = v[i];
        return;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)width;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (lane + 64 * i < nch) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)width + eps);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *yr = reinterpret_cast<float4 *>(y + row * (int64_t)width);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 ww = w4[c], bb = b4[c];
            yr[c] = make_float4((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                                (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
        }
    }
}

// Token assembly of the text tower (token embedding lookup + position embedding; chexzero_clip.py:363-365): one pass.
__global__ __launch_bounds__(256) void k_text_tokens(const int64_t *__restrict__ ids, int64_t ids_pitch, const float *__restrict__ tok,
                                                     const float *__restrict__ pos, int64_t batch, int seq, int width, int vocab,
                                                     float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= batch * seq) return;
    const int64_t bi = row / seq;
    const int t = (int)(row - bi * seq);
    int64_t id = ids[bi * ids_pitch + t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float4 *a = reinterpret_cast<const float4 *>(tok + id * (int64_t)width);
    const float4 *p = reinterpret_cast<const float4 *>(pos + (int64_t)t * width);
    float4 *o = reinterpret_cast<float4 *>(y + row * (int64_t)width);
    for (int c = lane; c < (width >> 2); c += 64) {
        const float4 u = a[c], v = p[c];
        o[c] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

}  // namespace

extern "C" int lemon_vision_tokens_ln(const float *patches_dev, const float *cls_dev, const float *pos_dev,
                                      const float *ln_weight_dev, const float *ln_bias_dev, float eps, int64_t batch,
                                      int n_tokens, int width, float *y_dev, void *stream_) {
    LEMON_REQUIRE(batch >= 0 && n_tokens >= 2 && width > 0 && (width & 3) == 0 && width <= 2048, "batch >= 0, n_tokens >= 2, width % 4 == 0, <= 2048");
    if (batch == 0) return LEMON_OK;
    LEMON_REQUIRE(patches_dev && cls_dev && pos_dev && y_dev, "null pointer");
    LEMON_REQUIRE((ln_weight_dev != nullptr) == (ln_bias_dev != nullptr), "LayerNorm weight and bias come together (both null: no LayerNorm)");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)((batch * n_tokens + 3) / 4)), block(256);
#define LAUNCH(CH) hipLaunchKernelGGL(k_vision_tokens_ln<CH>, grid, block, 0, stream, patches_dev, cls_dev, pos_dev, ln_weight_dev, \
                                      ln_bias_dev, eps, batch, n_tokens, width, y_dev)
    if (width <= 512) LAUNCH(2); else if (width <= 1024) LAUNCH(4); else LAUNCH(8);
#undef LAUNCH
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_text_tokens(const int64_t *ids_dev, int64_t ids_pitch, const float *tok_emb_dev, const float *pos_dev,
                                 int64_t batch, int seq_len, int width, int vocab, float *y_dev, void *stream_) {
    LEMON_REQUIRE(batch >= 0 && seq_len > 0 && width > 0 && (width & 3) == 0 && vocab > 0 && ids_pitch >= seq_len, "shapes");
    if (batch == 0) return LEMON_OK;
    LEMON_REQUIRE(ids_dev && tok_emb_dev && pos_dev && y_dev, "null pointer");
    hipLaunchKernelGGL(k_text_tokens, dim3((unsigned)((batch * seq_len + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, ids_dev,
                       ids_pitch, tok_emb_dev, pos_dev, batch, seq_len, width, vocab, y_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_layernorm_f32(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps,
                                   int64_t rows, int width, float *y_dev, void *stream_) {
    LEMON_REQUIRE(rows >= 0 && width > 0 && (width & 3) == 0 && width <= 2048, "rows >= 0, width a multiple of 4, <= 2048");
    if (rows == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev && weight_dev && bias_dev && y_dev, "null pointer");
    LEMON_REQUIRE(((((uintptr_t)x_dev) | ((uintptr_t)y_dev) | ((uintptr_t)weight_dev) | ((uintptr_t)bias_dev)) & 15) == 0,
                  "16-byte aligned pointers");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if ((width & 7) == 0) {
        if (width <= 512) hipLaunchKernelGGL((k_layernorm8<1, 0>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y_dev);
        else if (width <= 1024) hipLaunchKernelGGL((k_layernorm8<2, 0>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y_dev);
        else hipLaunchKernelGGL((k_layernorm8<4, 0>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y_dev);
    }
    else if (width <= 512) hipLaunchKernelGGL(k_layernorm<2>, grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y_dev);
    else if (width <= 1024) hipLaunchKernelGGL(k_layernorm<4>, grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y_dev);
    else hipLaunchKernelGGL(k_layernorm<8>, grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

// LayerNorm whose output is the split activation operand of lemon_linear_bf16x6 ([rows, 6 width] bf16) or lemon_linear_f16x3
// ([rows, 3 width] fp16): the same arithmetic as lemon_layernorm_f32 (the fp32 result is split, not recomputed), one pass.
template <int SCHEME>
static int layernorm_split_impl(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps,
                                int64_t rows, int width, uint16_t *y6_dev, void *stream_) {
    LEMON_REQUIRE(rows >= 0 && width > 0 && (width & 3) == 0 && width <= 2048, "rows >= 0, width a multiple of 4, <= 2048");
    if (rows == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev && weight_dev && bias_dev && y6_dev, "null pointer");
    LEMON_REQUIRE(((((uintptr_t)x_dev) | ((uintptr_t)weight_dev) | ((uintptr_t)bias_dev)) & 15) == 0 && (((uintptr_t)y6_dev) & 7) == 0,
                  "aligned pointers");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    float *y = reinterpret_cast<float *>(y6_dev);
    if ((width & 7) == 0 && (((uintptr_t)y6_dev) & 15) == 0) {
        if (width <= 512) hipLaunchKernelGGL((k_layernorm8<1, SCHEME>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y);
        else if (width <= 1024) hipLaunchKernelGGL((k_layernorm8<2, SCHEME>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y);
        else hipLaunchKernelGGL((k_layernorm8<4, SCHEME>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y);
    }
    else if (width <= 512) hipLaunchKernelGGL((k_layernorm<2, SCHEME>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y);
    else if (width <= 1024) hipLaunchKernelGGL((k_layernorm<4, SCHEME>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y);
    else hipLaunchKernelGGL((k_layernorm<8, SCHEME>), grid, block, 0, stream, x_dev, weight_dev, bias_dev, eps, rows, width, y);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
extern "C" int lemon_layernorm_split3(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps,
                                      int64_t rows, int width, uint16_t *y6_dev, void *stream_) {
    return layernorm_split_impl<1>(x_dev, weight_dev, bias_dev, eps, rows, width, y6_dev, stream_);
}
extern "C" int lemon_layernorm_f16x3(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps,
                                     int64_t rows, int width, uint16_t *y3_dev, void *stream_) {
    return layernorm_split_impl<2>(x_dev, weight_dev, bias_dev, eps, rows, width, y3_dev, stream_);
}

// ... as the tile-major fp16 operand of lemon_linear_f16x3t: yt_dev holds ceil(rows / 128) * 128 x width x 2 halves (rows beyond
// `rows` are not written; the GEMM never stores what it computes from them)
extern "C" int lemon_layernorm_f16x3t(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps,
                                      int64_t rows, int width, uint16_t *yt_dev, void *stream_) {
    LEMON_REQUIRE(rows >= 0 && width > 0 && (width & 15) == 0 && width <= 2048, "rows >= 0, width a multiple of 16, <= 2048");
    if (rows == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev && weight_dev && bias_dev && yt_dev, "null pointer");
    LEMON_REQUIRE(((((uintptr_t)x_dev) | ((uintptr_t)weight_dev) | ((uintptr_t)bias_dev) | ((uintptr_t)yt_dev)) & 15) == 0, "aligned pointers");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)((rows + 7) / 8)), block(256);
    const size_t lds = (size_t)(2 * (width / 8) + 1) * 8 * 16;         // 24.1 KB at width 768
    if (lds > 65536)      // (width 2048: 65 664 B; the attribute is per device, this shape is rare: set on every such call)
        LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_layernorm8_t<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (width <= 512) hipLaunchKernelGGL((k_layernorm8_t<1>), grid, block, lds, stream, x_dev, weight_dev, bias_dev, eps, rows, width, yt_dev);
    else if (width <= 1024) hipLaunchKernelGGL((k_layernorm8_t<2>), grid, block, lds, stream, x_dev, weight_dev, bias_dev, eps, rows, width, yt_dev);
    else hipLaunchKernelGGL((k_layernorm8_t<4>), grid, block, lds, stream, x_dev, weight_dev, bias_dev, eps, rows, width, yt_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

// The input side of a folded LayerNorm for a tensor no GEMM epilogue produced (the first block of a tower): x as the tile-major
// operand + the rows' (rstd, -mean rstd).  Same statistics arithmetic as lemon_layernorm_f32.
extern "C" int lemon_rowstats_f16x3t(const float *x_dev, float eps, int64_t rows, int width, uint16_t *yt_dev, float *row_aff_dev, void *stream_) {
    LEMON_REQUIRE(rows >= 0 && width > 0 && (width & 15) == 0 && width <= 2048, "rows >= 0, width a multiple of 16, <= 2048");
    if (rows == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev && yt_dev && row_aff_dev, "null pointer");
    LEMON_REQUIRE(((((uintptr_t)x_dev) | ((uintptr_t)yt_dev) | ((uintptr_t)row_aff_dev)) & 15) == 0, "aligned pointers");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)((rows + 7) / 8)), block(256);
    const size_t lds = (size_t)(2 * (width / 8) + 1) * 8 * 16;
    float2 *aff = reinterpret_cast<float2 *>(row_aff_dev);
    if (lds > 65536)
        LEMON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_layernorm8_t<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (width <= 512) hipLaunchKernelGGL((k_layernorm8_t<1, true>), grid, block, lds, stream, x_dev, nullptr, nullptr, eps, rows, width, yt_dev, aff);
    else if (width <= 1024) hipLaunchKernelGGL((k_layernorm8_t<2, true>), grid, block, lds, stream, x_dev, nullptr, nullptr, eps, rows, width, yt_dev, aff);
    else hipLaunchKernelGGL((k_layernorm8_t<4, true>), grid, block, lds, stream, x_dev, nullptr, nullptr, eps, rows, width, yt_dev, aff);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

template <int SCHEME>
static int split_rows_impl(const float *x_dev, int64_t rows, int k, int weight, float wscale, uint16_t *y_dev, void *stream_) {
    LEMON_REQUIRE(rows >= 0 && k > 0 && (k & 3) == 0, "rows >= 0, k a multiple of 4");
    if (rows == 0) return LEMON_OK;
    LEMON_REQUIRE(x_dev && y_dev && (((uintptr_t)x_dev) & 15) == 0 && (((uintptr_t)y_dev) & 7) == 0, "aligned pointers");
    const int64_t threads = rows * (k >> 2);
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    if (weight) hipLaunchKernelGGL((k_split3_rows<SCHEME, true>), grid, block, 0, (hipStream_t)stream_, x_dev, rows, k, y_dev, wscale);
    else hipLaunchKernelGGL((k_split3_rows<SCHEME, false>), grid, block, 0, (hipStream_t)stream_, x_dev, rows, k, y_dev, wscale);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
// 3-way bf16 split of a row-major fp32 matrix [rows, k] into the 6k-long operand rows of lemon_linear_bf16x6:
// weight = 0: activation layout [hi | hi | mid | hi | mid | lo]; weight = 1: weight layout [hi | mid | hi | lo | mid | hi].
extern "C" int lemon_split3_f32(const float *x_dev, int64_t rows, int k, int weight, uint16_t *y6_dev, void *stream_) {
    return split_rows_impl<1>(x_dev, rows, k, weight, 1.0f, y6_dev, stream_);
}
// 2-way fp16 split of a row-major fp32 matrix [rows, k] into the 3k-long operand rows of lemon_linear_f16x3:
// weight = 0: activation layout [hi | hi | lo 2^11] of x; weight = 1: weight layout [hi | lo | hi 2^-11] of x * wscale
// (wscale: a power of two, see lemon_linear_f16x3; ignored for activations).
extern "C" int lemon_split_f16x3(const float *x_dev, int64_t rows, int k, int weight, float wscale, uint16_t *y3_dev, void *stream_) {
    if (weight) {
        int e = 0;
        LEMON_REQUIRE(wscale > 0.0f && std::frexp(wscale, &e) == 0.5f, "wscale must be a power of two");
    }
    return split_rows_impl<2>(x_dev, rows, k, weight, wscale, y3_dev, stream_);
}
