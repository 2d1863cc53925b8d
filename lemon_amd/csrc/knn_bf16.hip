// knn_bf16.hip -- bf16 MFMA filter + exact fp32 re-rank (LEMON_ALGO_BF16_FILTER).
#include "common.hpp"
int lemon_search_f32(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                     int64_t *I_dev, hipStream_t stream);
int lemon_search_bf16(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                      int64_t *I_dev, hipStream_t stream) {
    // TODO(round 1, stage 2): not built yet; the exact fp32 scan is the only algorithm.
    return lemon_search_f32(idx, q_dev, nq, k, D_dev, I_dev, stream);
}
