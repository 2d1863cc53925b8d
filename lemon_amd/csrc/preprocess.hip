// Image preprocessing of the embedding stage on the GPU (gfx950 only):
//   generic_transform = Resize(224, BICUBIC on the uint8 PIL image) -> CenterCrop(224) -> ToTensor ->
//   Normalize(CLIP mean/std)                                lib/datasets/utils.py:159-170
// for batches of equally sized uint8 HWC images (CIFAR: 32x32 -> 224x224, the x49 up-sampling the
// reference spends 8 DataLoader workers on, run_lemon.py:129-131).  PIL's resampler is integer
// arithmetic: per output pixel a window of <= ksize taps with 22-bit fixed-point coefficients,
// accumulator 1<<21 + sum(u8 * k), result clip8(acc >> 22), horizontal pass first, then vertical on the
// uint8 intermediate.  The coefficient tables are built on the host by the same double-precision
// recipe (lemon_amd/data.py::pil_bicubic_tables, pinned against PIL itself in the tests) and already
// cropped to the S output rows/columns; this kernel is the two integer passes + the float epilogue
// (v/255 - mean)/std, bit-identical to the PIL + torch pipeline.
//
// One workgroup per (image, block of R output rows): horizontal pass for the input rows that block
// needs into an LDS uint8 tile, vertical pass from LDS (four pixels x three channels per work item: three
// aligned LDS words per tap, the float epilogue from a 768-entry table, 16-byte stores).  Traffic: the
// uint8 image in (3 KB for CIFAR), 12*S*S bytes out (602 KB): HBM-write bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "split3.hpp"

namespace {

constexpr int PIL_PRECISION_BITS = 22;   // 32 - 8 - 2 (Pillow Resample.c)

struct PreParams {
    const uint8_t *img;      // [B, H, W, 3]
    const int32_t *kk_h, *bnd_h, *kk_v, *bnd_v;   // [S, ks_h], [S, 2] (xmin, count), [S, ks_v], [S, 2]
    float *out;              // [B, 3, S, S]
    int H, W, S, ks_h, ks_v, R, blocks_per_img;
    int patch;               // 0: NCHW; P > 0: patch-major [B, (S/P)^2, 3*P*P] (what the patch-embedding GEMM reads)
    unsigned short *out_t;   // non-null (with P > 0): the same patch rows as the TILE-MAJOR fp16 split operand of lemon_linear_f16x3t
    float mean[3], stdv[3];
};

__device__ __forceinline__ uint8_t pil_clip8(int acc) {
    const int v = acc >> PIL_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

constexpr int LUT_BYTES = 3 * 256 * 4;     // (v/255 - mean[c]) / std[c] for every uint8 value and channel
constexpr int VTAB_MAX_INTS = 512;         // the block's vertical windows: R x (ymin, count, ks_v taps)

__global__ __launch_bounds__(256) void k_preprocess_u8(PreParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    float *s_lut = reinterpret_cast<float *>(s_dyn);                     // [3][256]
    int *s_vt = reinterpret_cast<int *>(s_dyn + LUT_BYTES);              // [R][2 + ks_v]
    uint8_t *s_tmp = s_dyn + LUT_BYTES + VTAB_MAX_INTS * 4;              // [rows][S*3] horizontally resampled input rows
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / p.blocks_per_img;
    const int yb = blockIdx.x % p.blocks_per_img;
    const int y0 = yb * p.R;
    const int y1 = y0 + p.R < p.S ? y0 + p.R : p.S;
    const int S = p.S, S3 = 3 * p.S;
    const int vmin = p.bnd_v[2 * y0];            // windows are monotone in y
    const int vmax = p.bnd_v[2 * (y1 - 1)] + p.bnd_v[2 * (y1 - 1) + 1];
    const int rows = vmax - vmin;
    const uint8_t *img = p.img + b * (int64_t)p.H * p.W * 3;

    // the float epilogue of a uint8 value is one of 768 numbers: ToTensor's division and Normalize's, done once per value with
    // exactly the arithmetic of the per-pixel form (two IEEE divisions per output pixel were most of this kernel's time)
    for (int i = tid; i < 768; i += blockDim.x) {
        const int c = i >> 8;
        const float f = (float)(i & 255) / 255.0f;                       // ToTensor
        s_lut[i] = (f - p.mean[c]) / p.stdv[c];                          // Normalize
    }
    const int vstride = 2 + p.ks_v;
    for (int i = tid; i < (y1 - y0) * vstride; i += blockDim.x) {
        const int yy = i / vstride, j = i - yy * vstride;
        s_vt[i] = j < 2 ? p.bnd_v[2 * (y0 + yy) + j] : p.kk_v[(y0 + yy) * p.ks_v + (j - 2)];
    }
    for (int r = 0; r < rows; ++r) {                                 // uniform outer loops: no per-element div/mod
        const uint8_t *row = img + (int64_t)(vmin + r) * p.W * 3;
        for (int xc = tid; xc < S3; xc += blockDim.x) {
            const int x = xc / 3, c = xc - 3 * x;
            const int xmin = p.bnd_h[2 * x], n = p.bnd_h[2 * x + 1];
            const int32_t *k = p.kk_h + x * p.ks_h;
            const uint8_t *src = row + xmin * 3 + c;
            int acc = 1 << (PIL_PRECISION_BITS - 1);
            for (int t = 0; t < n; ++t) acc += (int)src[3 * t] * k[t];
            s_tmp[r * S3 + xc] = pil_clip8(acc);
        }
    }
    __syncthreads();

    const int P = p.patch, nP = P ? S / P : 0;
    if ((S & 3) == 0 && (P & 3) == 0) {
        // four output pixels x three channels per work item: 12 bytes (3 aligned words) of the LDS tile per tap, three 16-byte stores
        const int q4 = S >> 2;
        for (int it = tid; it < (y1 - y0) * q4; it += blockDim.x) {
            const int yy = it / q4, x = 4 * (it - yy * q4), y = y0 + yy;
            const int *vt = s_vt + yy * vstride;
            const int ymin = vt[0], n = vt[1];
            int acc[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = 1 << (PIL_PRECISION_BITS - 1);
            const uint8_t *src = s_tmp + (ymin - vmin) * S3 + 3 * x;
            for (int t = 0; t < n; ++t) {
                const int kt = vt[2 + t];
                const uint32_t *w = reinterpret_cast<const uint32_t *>(src + t * S3);
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] += (int)((w0 >> (8 * i)) & 255u) * kt;
                    acc[4 + i] += (int)((w1 >> (8 * i)) & 255u) * kt;
                    acc[8 + i] += (int)((w2 >> (8 * i)) & 255u) * kt;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float *lut = s_lut + 256 * c;
                const float4 v = make_float4(lut[pil_clip8(acc[c])], lut[pil_clip8(acc[3 + c])], lut[pil_clip8(acc[6 + c])], lut[pil_clip8(acc[9 + c])]);
                if (p.out_t) {
                    // the patch-embedding GEMM's activation operand (split3.hpp: tiled_off): four consecutive k of one patch row = 8 bytes
                    // of its hi plane and 8 of its lo plane; no fp32 pixel tensor, no split pass
                    const float x4[4] = {v.x, v.y, v.z, v.w};
                    lemon_split::us4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned short a_, b_, c_;
                        lemon_split::split2h<false>(x4[e], a_, b_, c_);
                        hi[e] = a_; lo[e] = c_;
                    }
                    const int64_t o = lemon_split::tiled_off(lemon_split::TILE_A_ROWS, b * nP * nP + (int64_t)(y / P) * nP + x / P,
                                                             c * P * P + (y % P) * P + (x % P), 0, 3 * P * P);
                    *reinterpret_cast<lemon_split::us4 *>(p.out_t + o) = hi;
                    *reinterpret_cast<lemon_split::us4 *>(p.out_t + o + lemon_split::TILE_A_ROWS * 16) = lo;
                    continue;
                }
                // NCHW row, or the row's place inside its patches: out[b][py*nP+px][c*P*P + (y%P)*P + (x%P)]
                float *dst = P ? p.out + (b * nP * nP + (int64_t)(y / P) * nP + x / P) * (3 * P * P) + c * P * P + (y % P) * P + (x % P)
                               : p.out + ((b * 3 + c) * S + y) * (int64_t)S + x;
                *reinterpret_cast<float4 *>(dst) = v;
            }
        }
        return;
    }
    for (int c = 0; c < 3; ++c) {
        const float *lut = s_lut + 256 * c;
        for (int y = y0; y < y1; ++y) {
            const int *vt = s_vt + (y - y0) * vstride;
            const int ymin = vt[0], n = vt[1];
            float *dst = P ? p.out + (b * nP * nP + (int64_t)(y / P) * nP) * (3 * P * P) + c * P * P + (y % P) * P
                           : p.out + ((b * 3 + c) * S + y) * (int64_t)S;
            for (int x = tid; x < S; x += blockDim.x) {
                const uint8_t *src = s_tmp + (ymin - vmin) * S3 + 3 * x + c;
                int acc = 1 << (PIL_PRECISION_BITS - 1);
                for (int t = 0; t < n; ++t) acc += (int)src[t * S3] * vt[2 + t];
                const float v = lut[pil_clip8(acc)];
                if (P) dst[(int64_t)(x / P) * (3 * P * P) + (x % P)] = v; else dst[x] = v;
            }
        }
    }
}

}  // namespace

static int preprocess_impl(const uint8_t *img_dev, int64_t batch, int in_h, int in_w, const int32_t *kk_h_dev,
                           const int32_t *bnd_h_dev, int ks_h, const int32_t *kk_v_dev, const int32_t *bnd_v_dev,
                           int ks_v, int out_size, int max_rows_per_block, int rows_per_block,
                           const float *mean3_host, const float *std3_host, int patch, float *out_dev, unsigned short *out_t,
                           void *stream) {
    LEMON_REQUIRE(batch >= 0 && in_h > 0 && in_w > 0 && out_size > 0, "batch >= 0, sizes > 0");
    LEMON_REQUIRE(ks_h > 0 && ks_v > 0 && rows_per_block > 0 && max_rows_per_block > 0, "table geometry");
    LEMON_REQUIRE(patch >= 0 && (patch == 0 || out_size % patch == 0), "patch must divide out_size");
    if (batch == 0) return LEMON_OK;
    LEMON_REQUIRE(img_dev && kk_h_dev && bnd_h_dev && kk_v_dev && bnd_v_dev && mean3_host && std3_host && out_dev, "null pointer");
    LEMON_REQUIRE((int64_t)rows_per_block * (2 + ks_v) <= VTAB_MAX_INTS, "rows_per_block * (2 + ks_v) <= 512");
    LEMON_REQUIRE((((uintptr_t)out_dev) & 15) == 0, "out_dev must be 16-byte aligned");
    const size_t lds = LUT_BYTES + VTAB_MAX_INTS * 4 + (((size_t)max_rows_per_block * out_size * 3 + 15) & ~(size_t)15);
    LEMON_REQUIRE((size_t)max_rows_per_block * out_size * 3 <= 56 * 1024, "rows_per_block too large: the horizontal tile must fit 56 KB of LDS");
    PreParams p;
    p.img = img_dev; p.kk_h = kk_h_dev; p.bnd_h = bnd_h_dev; p.kk_v = kk_v_dev; p.bnd_v = bnd_v_dev; p.out = out_dev;
    p.patch = patch; p.out_t = out_t;
    p.H = in_h; p.W = in_w; p.S = out_size; p.ks_h = ks_h; p.ks_v = ks_v; p.R = rows_per_block;
    p.blocks_per_img = (out_size + rows_per_block - 1) / rows_per_block;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3_host[c]; p.stdv[c] = std3_host[c]; }
    const int64_t grid = batch * p.blocks_per_img;
    LEMON_REQUIRE(grid < (int64_t)1 << 31, "batch * row blocks < 2^31");
    hipLaunchKernelGGL(k_preprocess_u8, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

extern "C" int lemon_preprocess_u8(const uint8_t *img_dev, int64_t batch, int in_h, int in_w, const int32_t *kk_h_dev,
                                   const int32_t *bnd_h_dev, int ks_h, const int32_t *kk_v_dev, const int32_t *bnd_v_dev,
                                   int ks_v, int out_size, int max_rows_per_block, int rows_per_block,
                                   const float *mean3_host, const float *std3_host, int patch, float *out_dev,
                                   void *stream) {
    return preprocess_impl(img_dev, batch, in_h, in_w, kk_h_dev, bnd_h_dev, ks_h, kk_v_dev, bnd_v_dev, ks_v, out_size, max_rows_per_block,
                           rows_per_block, mean3_host, std3_host, patch, out_dev, nullptr, stream);
}

// ... with the patch rows written as the tile-major fp16 split operand of lemon_linear_f16x3t (rows = batch * (out_size / patch)^2,
// k = 3 patch^2): the patch embedding then runs in the hand-written GEMM straight from this kernel's output
extern "C" int lemon_preprocess_u8_f16x3t(const uint8_t *img_dev, int64_t batch, int in_h, int in_w, const int32_t *kk_h_dev,
                                          const int32_t *bnd_h_dev, int ks_h, const int32_t *kk_v_dev, const int32_t *bnd_v_dev,
                                          int ks_v, int out_size, int max_rows_per_block, int rows_per_block,
                                          const float *mean3_host, const float *std3_host, int patch, uint16_t *outt_dev,
                                          void *stream) {
    LEMON_REQUIRE(patch > 0 && patch % 4 == 0 && out_size % 4 == 0 && (3 * patch * patch) % 16 == 0,
                  "patch a positive multiple of 4 with 3 patch^2 a multiple of 16 (the operand's k16 steps)");
    LEMON_REQUIRE(outt_dev != nullptr, "null pointer");
    return preprocess_impl(img_dev, batch, in_h, in_w, kk_h_dev, bnd_h_dev, ks_h, kk_v_dev, bnd_v_dev, ks_v, out_size, max_rows_per_block,
                           rows_per_block, mean3_host, std3_host, patch, reinterpret_cast<float *>(outt_dev), outt_dev, stream);
}
