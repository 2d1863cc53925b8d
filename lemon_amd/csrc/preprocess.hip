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
// needs into an LDS uint8 tile, vertical pass from LDS, coalesced float stores (NCHW).  Traffic: the
// uint8 image in (3 KB for CIFAR), 12*S*S bytes out (602 KB): HBM-write bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"

namespace {

constexpr int PIL_PRECISION_BITS = 22;   // 32 - 8 - 2 (Pillow Resample.c)

struct PreParams {
    const uint8_t *img;      // [B, H, W, 3]
    const int32_t *kk_h, *bnd_h, *kk_v, *bnd_v;   // [S, ks_h], [S, 2] (xmin, count), [S, ks_v], [S, 2]
    float *out;              // [B, 3, S, S]
    int H, W, S, ks_h, ks_v, R, blocks_per_img;
    int patch;               // 0: NCHW; P > 0: patch-major [B, (S/P)^2, 3*P*P] (what the patch-embedding GEMM reads)
    float mean[3], stdv[3];
};

__device__ __forceinline__ uint8_t pil_clip8(int acc) {
    const int v = acc >> PIL_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void k_preprocess_u8(PreParams p) {
    extern __shared__ uint8_t s_tmp[];           // [rows][S*3] horizontally resampled input rows
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

    for (int c = 0; c < 3; ++c) {
        const float mean = p.mean[c], stdv = p.stdv[c];
        for (int y = y0; y < y1; ++y) {
            const int ymin = p.bnd_v[2 * y], n = p.bnd_v[2 * y + 1];
            const int32_t *k = p.kk_v + y * p.ks_v;
            // NCHW row, or the row's place inside its patches: out[b][py*nP+px][c*P*P + (y%P)*P + (x%P)]
            const int P = p.patch, nP = P ? S / P : 0;
            float *dst = P ? p.out + (b * nP * nP + (int64_t)(y / P) * nP) * (3 * P * P) + c * P * P + (y % P) * P
                           : p.out + ((b * 3 + c) * S + y) * (int64_t)S;
            for (int x = tid; x < S; x += blockDim.x) {
                const uint8_t *src = s_tmp + (ymin - vmin) * S3 + 3 * x + c;
                int acc = 1 << (PIL_PRECISION_BITS - 1);
                for (int t = 0; t < n; ++t) acc += (int)src[t * S3] * k[t];
                float f = (float)pil_clip8(acc) / 255.0f;                // ToTensor
                const float v = (f - mean) / stdv;                       // Normalize
                if (P) dst[(int64_t)(x / P) * (3 * P * P) + (x % P)] = v; else dst[x] = v;
            }
        }
    }
}

}  // namespace

extern "C" int lemon_preprocess_u8(const uint8_t *img_dev, int64_t batch, int in_h, int in_w, const int32_t *kk_h_dev,
                                   const int32_t *bnd_h_dev, int ks_h, const int32_t *kk_v_dev, const int32_t *bnd_v_dev,
                                   int ks_v, int out_size, int max_rows_per_block, int rows_per_block,
                                   const float *mean3_host, const float *std3_host, int patch, float *out_dev,
                                   void *stream) {
    LEMON_REQUIRE(batch >= 0 && in_h > 0 && in_w > 0 && out_size > 0, "batch >= 0, sizes > 0");
    LEMON_REQUIRE(ks_h > 0 && ks_v > 0 && rows_per_block > 0 && max_rows_per_block > 0, "table geometry");
    LEMON_REQUIRE(patch >= 0 && (patch == 0 || out_size % patch == 0), "patch must divide out_size");
    if (batch == 0) return LEMON_OK;
    LEMON_REQUIRE(img_dev && kk_h_dev && bnd_h_dev && kk_v_dev && bnd_v_dev && mean3_host && std3_host && out_dev, "null pointer");
    const size_t lds = (size_t)max_rows_per_block * out_size * 3;
    LEMON_REQUIRE(lds <= 64 * 1024, "rows_per_block too large: the horizontal tile must fit 64 KB of LDS");
    PreParams p;
    p.img = img_dev; p.kk_h = kk_h_dev; p.bnd_h = bnd_h_dev; p.kk_v = kk_v_dev; p.bnd_v = bnd_v_dev; p.out = out_dev;
    p.patch = patch;
    p.H = in_h; p.W = in_w; p.S = out_size; p.ks_h = ks_h; p.ks_v = ks_v; p.R = rows_per_block;
    p.blocks_per_img = (out_size + rows_per_block - 1) / rows_per_block;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3_host[c]; p.stdv[c] = std3_host[c]; }
    const int64_t grid = batch * p.blocks_per_img;
    LEMON_REQUIRE(grid < (int64_t)1 << 31, "batch * row blocks < 2^31");
    hipLaunchKernelGGL(k_preprocess_u8, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
