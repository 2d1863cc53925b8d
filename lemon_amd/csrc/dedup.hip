// Query de-duplication for the flat search (gfx950).
//
// On classification datasets the text-side queries of run_lemon.py:236 are N copies of C class-prompt embeddings
// (50 000 queries, 100 distinct rows on CIFAR-100).  Identical query rows have identical results, so
// the search runs once per DISTINCT row and the (D, I) lists are copied to every member: exact by construction (no
// tie-order subtlety: a query's result does not depend on the other queries).  Grouping is content based -- 64-bit row
// hash, stable radix sort, then a full bitwise comparison of each row with its predecessor in sorted order, so a hash
// collision can only split a group, never merge two different rows.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace {

__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// one wavefront per row: position-salted mix of every element's bits, xor/add-combined across lanes
__global__ __launch_bounds__(256) void k_row_hash(const float *__restrict__ q, int64_t n, int d, u64 *__restrict__ h,
                                                  u32 *__restrict__ iota) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const u32 *r = reinterpret_cast<const u32 *>(q + row * d);
    u64 acc = 0;
    for (int c = lane; c < d; c += 64) acc += mix64(((u64)(u32)c << 32) | (u64)r[c]);
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) { h[row] = mix64(acc); iota[row] = (u32)row; }
}

// head[i] = 1 when sorted position i starts a new group (first, different hash, or different content)
__global__ __launch_bounds__(256) void k_group_heads(const float *__restrict__ q, int d, int64_t n, const u64 *__restrict__ hs,
                                                     const u32 *__restrict__ order, int *__restrict__ head) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    int differ = (i == 0) || (hs[i] != hs[i - 1]);
    if (!differ) {
        const u32 *a = reinterpret_cast<const u32 *>(q + (int64_t)order[i] * d);
        const u32 *b = reinterpret_cast<const u32 *>(q + (int64_t)order[i - 1] * d);
        int diff = 0;
        for (int c = lane; c < d; c += 64) diff |= (a[c] != b[c]);
        differ = __any(diff) ? 1 : 0;
    }
    if (lane == 0) head[i] = differ;
}

// gid = inclusive_scan(head) - 1; representative of a group = its first member in sorted order = its LOWEST original
// row (the sort is stable and starts from ascending row numbers)
__global__ void k_assign_groups(const u32 *__restrict__ order, const int *__restrict__ head, const int *__restrict__ scan,
                                int64_t n, int *__restrict__ rep, int *__restrict__ group_of) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = scan[i] - 1;
    group_of[order[i]] = g;
    if (head[i]) rep[g] = (int)order[i];
}

__global__ void k_gather_rows(const float *__restrict__ q, const int *__restrict__ rep, int64_t U, int d, float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = U * d;
    if (t >= total) return;
    const int64_t g = t / d;
    out[t] = q[(int64_t)rep[g] * d + (t - g * d)];
}

__global__ void k_expand_results(const float *__restrict__ Dr, const int64_t *__restrict__ Ir, const int *__restrict__ group_of,
                                 int64_t nq, int k, float *__restrict__ D, int64_t *__restrict__ I) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * k) return;
    const int64_t j = t / k;
    const int64_t src = (int64_t)group_of[j] * k + (t - j * k);
    D[t] = Dr[src];
    I[t] = Ir[src];
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// Groups the rows of q [nq, d]: *U_host distinct rows (SYNCHRONISES the stream once to read the count), rep_dev[U] =
// lowest row of each group, group_dev[nq] = group of every row.  The arrays live in the index's dedup workspace and stay
// valid until the next call on the same handle.
int lemon_dedup_queries(lemon_index_t *idx, const float *q_dev, int64_t nq, hipStream_t stream, int64_t *U_host,
                        const int **rep_dev, const int **group_dev) {
    const int d = idx->d;
    size_t t1 = 0, t2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const u64 *)nullptr, (u64 *)nullptr, (const u32 *)nullptr, (u32 *)nullptr,
                                             (int)nq, 0, 64, stream);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, t2, (const int *)nullptr, (int *)nullptr, (int)nq, stream);
    const size_t tmp = align256(t1 > t2 ? t1 : t2);
    const size_t n8 = align256((size_t)nq * 8), n4 = align256((size_t)nq * 4);
    const size_t need = 2 * n8 + 6 * n4 + tmp + 256;
    if (need > idx->ws_dd_bytes) {
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (idx->ws_dd) (void)hipFree(idx->ws_dd);
        idx->ws_dd = nullptr; idx->ws_dd_bytes = 0;
        if (hipMalloc(&idx->ws_dd, need) != hipSuccess) { lemon_set_error("dedup workspace allocation failed (%zu bytes)", need); return LEMON_E_NOMEM; }
        idx->ws_dd_bytes = need;
    }
    char *p = (char *)idx->ws_dd;
    u64 *h = (u64 *)p; p += n8;
    u64 *hs = (u64 *)p; p += n8;
    u32 *iota = (u32 *)p; p += n4;
    u32 *order = (u32 *)p; p += n4;
    int *head = (int *)p; p += n4;
    int *scan = (int *)p; p += n4;
    int *rep = (int *)p; p += n4;
    int *group_of = (int *)p; p += n4;
    void *cub_tmp = p;
    const unsigned rows4 = (unsigned)((nq + 3) / 4);
    hipLaunchKernelGGL(k_row_hash, dim3(rows4), dim3(256), 0, stream, q_dev, nq, d, h, iota);
    LEMON_HIP_CHECK(hipGetLastError());
    size_t tb = tmp;
    if (hipcub::DeviceRadixSort::SortPairs(cub_tmp, tb, h, hs, iota, order, (int)nq, 0, 64, stream) != hipSuccess) {
        lemon_set_error("hipcub radix sort failed"); return LEMON_E_HIP;
    }
    hipLaunchKernelGGL(k_group_heads, dim3(rows4), dim3(256), 0, stream, q_dev, d, nq, hs, order, head);
    LEMON_HIP_CHECK(hipGetLastError());
    tb = tmp;
    if (hipcub::DeviceScan::InclusiveSum(cub_tmp, tb, head, scan, (int)nq, stream) != hipSuccess) {
        lemon_set_error("hipcub scan failed"); return LEMON_E_HIP;
    }
    hipLaunchKernelGGL(k_assign_groups, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, order, head, scan, nq, rep, group_of);
    LEMON_HIP_CHECK(hipGetLastError());
    int U = 0;
    LEMON_HIP_CHECK(hipMemcpyAsync(&U, scan + (nq - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    LEMON_HIP_CHECK(hipStreamSynchronize(stream));
    *U_host = U;
    *rep_dev = rep;
    *group_dev = group_of;
    return LEMON_OK;
}

int lemon_gather_query_rows(const float *q_dev, const int *rep_dev, int64_t U, int d, float *out_dev, hipStream_t stream) {
    const int64_t total = U * d;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, q_dev, rep_dev, U, d, out_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

int lemon_expand_results(const float *Dr, const int64_t *Ir, const int *group_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                         hipStream_t stream) {
    const int64_t total = nq * k;
    hipLaunchKernelGGL(k_expand_results, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, Dr, Ir, group_dev, nq, k, D_dev, I_dev);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}
