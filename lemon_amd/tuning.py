"""GEMM solution selection for the encoder (plumbing around hipBLASLt, not a kernel).

hipBLASLt's default heuristic picks a 64x64x128 macro-tile for the ViT-B/32 QKV projection
(50 000 x 768 x 2304, fp32) that runs at ~120 TFLOP/s; benchmarking its own solution list finds one at
~142.  PyTorch's TunableOp does that benchmarking; `data/tunableop_gfx950.csv` holds the winners for the
shapes of the headline workload (tools/tune_gemms.py regenerates it on an MI355X), shapes not in the
file are tuned on first use (bounded).  Results are keyed by library versions: a mismatch just re-tunes.
"""
import os
import shutil
import tempfile

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tunableop_gfx950.csv")


def enable_gemm_tuning(tune_missing=True, max_ms=1500, max_iters=30, results=RESULTS):
    """Turn TunableOp on for this process.  The committed results are copied to a per-process scratch
    file first (TunableOp appends what it tunes; ranks must not race on a shared file)."""
    import torch.cuda.tunable as tn
    work = os.path.join(tempfile.gettempdir(), f"lemon_tunableop_{os.getpid()}.csv")
    if results and os.path.exists(results):
        shutil.copyfile(results, work)
    elif os.path.exists(work):
        os.remove(work)
    tn.enable(True)
    tn.set_filename(work)
    tn.tuning_enable(bool(tune_missing))
    tn.set_max_tuning_duration(int(max_ms))
    tn.set_max_tuning_iterations(int(max_iters))
    return work
