"""lemon_amd -- MI355X-native implementation of LEMoN's embed -> kNN -> score hot path.

Only what the path needs lives here: csrc/ (HIP kernels + C ABI, include/lemon_hip.h), the
host-side mirror of the reference's interface (index / ops / neighbors / models / run_lemon).
"""
from ._lib import LemonHipError, METRIC_IP, METRIC_L2, ALGO_AUTO, ALGO_F32_MFMA, ALGO_BF16_FILTER  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require a GPU (the CPU test-suite imports host logic)
    if name in ("IndexFlatIP", "IndexFlatL2"):
        from . import index
        return getattr(index, name)
    if name in ("normalize_vectors", "paired_distance", "d1_normalized", "lemon_score",
                "calc_scores_given_hparams_vectorized", "calc_scores_given_hparams"):
        from . import ops
        return getattr(ops, name)
    if name == "LemonDB":
        from . import neighbors
        return neighbors.LemonDB
    raise AttributeError(name)
