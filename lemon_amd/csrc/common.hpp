// common.hpp -- shared host/device helpers of liblemon_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <float.h>
#include "../../include/lemon_hip.h"
#include <vector>
#include <utility>

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- error plumbing -------------------------------------------------------------
void lemon_set_error(const char *fmt, ...);

#define LEMON_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            lemon_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LEMON_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define LEMON_REQUIRE(cond, msg)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            lemon_set_error("invalid argument: %s (%s)", msg, #cond);                      \
            return LEMON_E_INVALID;                                                        \
        }                                                                                  \
    } while (0)

// ---- index object ---------------------------------------------------------------
// Data layout in HBM (DESIGN.md "Data layout"):
//   x     [cap, d]        row-major copy of what the caller added (gathers, exact re-rank)
//   xp    [cap_pad, dpad] same rows, zero padded to dpad = ceil64(d) columns and with every
//                         group of 8 consecutive k stored as [k0 k2 k4 k6 | k1 k3 k5 k7] so
//                         that one 16-B LDS read feeds four v_mfma_f32_32x32x2_f32 steps in
//                         ascending-k order (cap_pad = ceil128(cap), pad rows are zero)
//   xnorm [cap_pad]       dot(x,x) (chain numerics), used by the L2 epilogue
struct lemon_index {
    int metric;
    int d, dpad;
    int device;
    int algo;
    int64_t n, cap;       // rows stored / rows allocated (cap is a multiple of 128)
    float *x, *xp, *xnorm;
    // bf16 filter copies (built lazily by the bf16 path)
    unsigned short *xh;   // [cap, dpad_h] bf16 (RNE) of x
    float *xh_stats;      // [2, cap]: measured ||x-xh||^2 and ||xh||^2 per row
    int64_t xh_rows;      // rows of xh that are up to date
    unsigned *xn2max_dev; // device scalars (float bits): max dot(x,x), max ||x-xh||^2, max ||xh||^2
    // search workspace (grown on demand)
    int64_t ws_q;         // query rows the workspace is sized for
    int64_t ws_qp_row_bytes;  // bytes per staged query row the workspace is sized for
    int dpad_h;           // column pitch of xh (0 until the bf16 copy exists)
    float *ws_qp;         // [ws_q, dpad] permuted queries
    float *ws_qnorm;      // [ws_q]
    u64 *ws_cand;         // [ws_cand_rows, CAND_CAP]  (one 128-row region per scan workgroup)
    int64_t ws_cand_rows;
    float *ws_state;      // [grid*128][2] per-query scan state carried across database chunks
    int64_t ws_state_elems;
    u64 *ws_part;         // [splits_cap, ws_q, LEMON_MAX_K]
    int64_t ws_part_elems;
    // fp32 scan: segment plans on the device (knn_f32.hip), a few recent (panels, tiles) shapes -- a pipeline alternates
    // between its train / val / test query counts.  `host` keeps the upload's source alive (stream-ordered copy).
    struct PlanSlot {
        int panels, tiles, grid, splits, pieces_off, segs_off;
        int *dev; int64_t ints; unsigned long long stamp; std::vector<int> *host;
    } plan_slots[4];
    unsigned long long plan_clock;
    // neighbours workspace
    int64_t ws_nb;        // elements
    float *ws_D;          // [ws_nb]
    int64_t *ws_I;        // [ws_nb]
    lemon_search_info_t last;
    // LEMON_ALGO_AUTO decision cache (valid while auto_n == n)
    int auto_algo;
    int64_t auto_n;
    // query de-duplication (dedup.hip): mode (0 off, 1 auto) and its workspace
    int qdedup;
    void *ws_dd;
    size_t ws_dd_bytes;
    float *ws_ddq;        // gathered representative queries + their results
    size_t ws_ddq_bytes;
    // optional scan-kernel timing (lemon_index_set_profiling)
    int profiling;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> *prof_events;
    double prof_flops, prof_bytes;
};

// bracket a kernel launch with events when profiling is on
struct LemonProfScope {
    lemon_index *idx; hipStream_t s; hipEvent_t a, b; bool on;
    LemonProfScope(lemon_index *i, hipStream_t st, double flops, double bytes) : idx(i), s(st), on(false) {
        if (!i->profiling) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        on = true;
        (void)hipEventRecord(a, s);
        i->prof_flops += flops; i->prof_bytes += bytes;
    }
    ~LemonProfScope() {
        if (!on) return;
        (void)hipEventRecord(b, s);
        idx->prof_events->push_back(std::make_pair(a, b));
    }
};

// ---- key packing: bigger key == better candidate ----------------------------------
// hi 32 bits: order-preserving map of the float score (to MAXIMISE), lo 32 bits: ~index so
// that, at equal score, the LOWER database index wins.  key 0 is "no candidate".
__host__ __device__ inline u32 lemon_f2ord(float f) {
    u32 u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float lemon_ord2f(u32 o) {
    u32 u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}
__host__ __device__ inline u64 lemon_make_key(float score, u32 idx) {
    return ((u64)lemon_f2ord(score) << 32) | (u64)(0xffffffffu - idx);
}
__host__ __device__ inline float lemon_key_score(u64 key) { return lemon_ord2f((u32)(key >> 32)); }
__host__ __device__ inline u32 lemon_key_index(u64 key) { return 0xffffffffu - (u32)(key & 0xffffffffu); }

#define LEMON_CAND_CAP 256 // candidate slots per query in the scan workspace
// LayerNorm fold (lemon_linear_f16x3t_ln): the folded GEMM's rounding error grows with sqrt(1 + mean^2 / var) of a row; rows
// beyond |mean| rstd = this get a NaN row affine -> a non-finite output row -> the caller's range fallback re-embeds the
// micro-batch with LayerNorm kernels (pipeline.Embedder), so the fold never silently loses accuracy
#define LEMON_LN_FOLD_MAX_SHIFT 8.0f
#define LEMON_DEDUP_MIN_NQ 1024 // query de-duplication is attempted from this many queries on

// internal entry points shared between translation units
int lemon_search_internal(lemon_index_t *idx, const float *q_dev, int64_t nq, int k,
                          float *D_dev, int64_t *I_dev, hipStream_t stream);
int lemon_rowdot_chain(const float *a, const float *b, int64_t n, int d, float *out, hipStream_t s);
int lemon_dedup_queries(lemon_index_t *idx, const float *q_dev, int64_t nq, hipStream_t stream, int64_t *U_host,
                        const int **rep_dev, const int **group_dev);
int lemon_gather_query_rows(const float *q_dev, const int *rep_dev, int64_t U, int d, float *out_dev, hipStream_t stream);
int lemon_expand_results(const float *Dr, const int64_t *Ir, const int *group_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                         hipStream_t stream);
