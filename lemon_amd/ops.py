"""Host-side mirror of the reference's row-wise helpers, running on liblemon_hip.so.

  normalize_vectors      lib/utils/utils.py:39-40
  paired_distance        run_lemon.py:169,173,250-253
  d1_normalized          run_lemon.py:244-248
  calc_scores_given_hparams(_vectorized)   lib/metrics/utils.py:21-82

Inputs are torch CUDA tensors (device memory is torch's job; the arithmetic is the HIP
library's).  Host inputs are rejected loudly: there is no CPU path in the product.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_METRICS = {"cosine": _lib.METRIC_IP, "ip": _lib.METRIC_IP, "euclidean": _lib.METRIC_L2,
            "l2": _lib.METRIC_L2, _lib.METRIC_IP: _lib.METRIC_IP, _lib.METRIC_L2: _lib.METRIC_L2}


def metric_id(metric):
    try:
        return _METRICS[metric]
    except KeyError:
        raise ValueError(f"unknown metric {metric!r} (cosine|euclidean)")


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dev_f32(t, name="tensor"):
    """Contiguous float32 CUDA tensor (converted if needed); refuses CPU tensors."""
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a torch CUDA tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _lib.LemonHipError(f"{name} lives on {t.device}: the LEMoN hot path only runs on the GPU "
                                 "(no CPU fallback). Move it with .cuda().")
    return t.detach().to(torch.float32).contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def normalize_vectors(vectors):
    """F.normalize(vectors, p=2, dim=1) -- lib/utils/utils.py:39-40."""
    x = dev_f32(vectors, "vectors")
    assert x.dim() == 2
    y = torch.empty_like(x)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.lemon_normalize_rows(ptr(x), x.shape[0], x.shape[1], ptr(y), stream_ptr(x.device)),
                   "lemon_normalize_rows")
    return y


def our_metric(first, second, dist="cosine"):
    """DistanceEvaluator.our_metric (lib/metrics/distance_metrics.py:48-73): paired cosine /
    euclidean (not squared) / manhattan distance of un-normalised embeddings, without the reference's
    n x n detour.  Returns a CUDA float32 tensor [n]."""
    kind = {"cosine": 0, "euclidean": 1, "manhattan": 2}.get(dist)
    if kind is None:
        raise NotImplementedError(dist)
    a, b = dev_f32(first, "first"), dev_f32(second, "second")
    assert a.shape == b.shape and a.dim() == 2
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    lib = _lib.load()
    with torch.cuda.device(a.device):
        _lib.check(lib.lemon_paired_metric(kind, ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(out),
                                           stream_ptr(a.device)), "lemon_paired_metric")
    return out


def grid_f1(rec, y, hparams, xtol=1e-8, maxfun=500, return_scores=False):
    """Batched hyper-parameter grid: for every row (beta, gamma, tau_1_n, tau_2_n, tau_1_m, tau_2_m) of
    `hparams` [G,6] the F1-optimal threshold search of optimize_f1_efficient on the device-resident
    record `rec` (lemon_grid_f1) -> (f1 [G], thres [G]) numpy float64, bit-identical to the host loop."""
    import numpy as np
    dev = rec["D_n"].device
    f32 = lambda key: dev_f32(rec[key], key)
    d1, Dn, trn, dn, Dm, trm, dm = (f32(key) for key in ("d_1", "D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m"))
    n, k = Dn.shape
    yb = torch.as_tensor(np.asarray(y).astype(bool).astype(np.uint8)).to(dev)
    hp = torch.as_tensor(np.asarray(hparams, dtype=np.float64).reshape(-1, 6)).to(dev)
    G = hp.shape[0]
    f1 = torch.empty(G, dtype=torch.float64, device=dev)
    th = torch.empty(G, dtype=torch.float64, device=dev)
    lib = _lib.load()
    chunk = max(1, min(65535, (1 << 30) // (8 * n)))             # <= 1 GiB of score workspace per launch
    ws = torch.empty((min(G, chunk), n), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        for g0 in range(0, G, chunk):
            g1 = min(G, g0 + chunk)
            _lib.check(lib.lemon_grid_f1(ptr(d1), ptr(Dn), ptr(trn), ptr(dn), ptr(Dm), ptr(trm), ptr(dm), ptr(yb), n, k,
                                         ptr(hp[g0:g1]), g1 - g0, float(xtol), int(maxfun), ptr(ws), ptr(f1[g0:g1]),
                                         ptr(th[g0:g1]), stream_ptr(dev)), "lemon_grid_f1")
    if return_scores:
        return f1.cpu().numpy(), th.cpu().numpy(), ws[:min(G, chunk)].cpu().numpy()
    return f1.cpu().numpy(), th.cpu().numpy()


ATTENTION_MAX_SEQ = 288
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
_ACT_CODE = {None: ACT_NONE, "silu": ACT_SILU, "gelu": ACT_GELU}      # 'gelu': the exact (erf) GELU of BiomedCLIP's towers
QUICK_GELU_SCALE = 1.702
_linear_tuned_loaded = set()      # device indices whose per-device hipBLASLt state has the recorded choices


def _ensure_linear_tuned(lib, device):
    """Hand the recorded hipBLASLt solution choices (lemon_amd/data/linear_gfx950.csv, or $LEMON_LINEAR_TUNED) to the
    library once per DEVICE: the library keeps one state per device (selected by hipGetDevice), so the file is loaded with
    `device` current -- a rank on cuda:3 gets the same solutions (same embedding bits, same speed) as rank 0.  A key that is
    not in the file uses the library's first-ranked solution (no timing; lemon_linear_set_tuning(1) turns timing on)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _linear_tuned_loaded:
        return
    _linear_tuned_loaded.add(idx)
    import os
    path = os.environ.get("LEMON_LINEAR_TUNED",
                          os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "linear_gfx950.csv"))
    if path and os.path.exists(path):
        with torch.cuda.device(device):
            lib.lemon_linear_load_tuned(path.encode())


def linear(x, weight, bias=None, residual=None, act=None, alpha=1.0):
    """y = act(alpha * x @ weight.T + bias) (+ residual) in ONE hipBLASLt GEMM (SiLU / residual add ride
    in the epilogue; 'gelu', the exact one, is an in-place pass behind it).  x [..., k] float32 CUDA, weight [n, k]; act in
    (None, 'silu', 'gelu')."""
    assert x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
    x = x.contiguous()
    weight = weight.contiguous()
    k = x.shape[-1]
    n = weight.shape[0]
    assert weight.shape[1] == k
    m = x.numel() // k
    y = torch.empty(x.shape[:-1] + (n,), dtype=torch.float32, device=x.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == torch.float32
        residual = residual.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    lib = _lib.load()
    _ensure_linear_tuned(lib, x.device)
    code = _ACT_CODE[act]
    with torch.cuda.device(x.device):
        _lib.check(lib.lemon_linear_f32(ptr(x), ptr(weight), ptr(bias) if bias is not None else None,
                                        ptr(residual) if residual is not None else None, m, n, k, float(alpha), code, ptr(y),
                                        stream_ptr(x.device)), "lemon_linear_f32")
    return y


# ---- fp32-equivalent GEMMs on the 16-bit matrix cores (split operands: lemon_linear_bf16x6 / lemon_linear_f16x3) -----------
_SCHEMES = {"bf16x6": (torch.bfloat16, 6), "f16x3": (torch.float16, 3)}


_forced_gemm_mode = None          # set by gemm_mode_forced(): overrides $LEMON_GEMM for the calls inside the context


class gemm_mode_forced:
    """with gemm_mode_forced("bf16x6"): ... -- the GEMM mode of the towers for the calls inside, whatever $LEMON_GEMM says
    (pipeline.Embedder re-runs a micro-batch whose fp16 split operands overflowed with the range-free bf16 scheme)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        global _forced_gemm_mode
        self.prev, _forced_gemm_mode = _forced_gemm_mode, self.mode
        return self

    def __exit__(self, *exc):
        global _forced_gemm_mode
        _forced_gemm_mode = self.prev
        return False


_attn_f16_state = None            # (kept for callers that reset it; the library is asked every time now)


def select_attention_arithmetic(mode):
    """The attention kernels' arithmetic follows the GEMM mode: split products on the fp16 matrix cores with 'f16x3' (inputs
    must stay inside the fp16 range, like the GEMM operands of that mode), v_mfma_f32_32x32x2_f32 with 'bf16x6' (the mode whose
    point is to have no range limit) and 'f32'.  $LEMON_ATTN_F16=0 keeps the fp32 form in every mode.  The library keeps the
    selection per calling thread (lemon_attention_set_f16) and this is called in front of every tower pass WITHOUT a host-side
    cache: a direct lemon_attention_set_f16 call, another embedder or another thread cannot leave a stale choice behind."""
    global _attn_f16_state
    import os
    want = 1 if (mode == "f16x3" and os.environ.get("LEMON_ATTN_F16", "1") != "0") else 0
    _lib.load().lemon_attention_set_f16(want)
    _attn_f16_state = want


def gemm_mode():
    """How the four GEMMs of every transformer block run (LEMON_GEMM):
    'f16x3' (default): 2-way fp16 split operands, three cross products, one fp16 GEMM over 3k -- the fp32 GEMM's accuracy
        (max / rms error vs float64 measured slightly below it), 2.4x faster than the tuned fp32 GEMMs at the tower shapes;
    'bf16x6' (alias 'split'): 3-way bf16 split operands, six cross products, one bf16 GEMM over 6k (same delivered accuracy --
        fp32 accumulation bounds both --, no fp16 range limit on the operands, 1.4x faster than fp32);
    'f32': every GEMM on the fp32 matrix cores."""
    import os
    v = (_forced_gemm_mode or os.environ.get("LEMON_GEMM", DEFAULT_GEMM_MODE)).lower()
    if v in ("f32", "fp32", "0"):
        return "f32"
    if v in ("split", "bf16x6"):
        return "bf16x6"
    if v in ("f16x3", "fp16x3"):
        return "f16x3"
    raise ValueError(f"LEMON_GEMM={v!r}: expected f32, bf16x6 (split) or f16x3")


DEFAULT_GEMM_MODE = "f16x3"


def weight_scale_f16x3(w):
    """The power of two that lifts max|w| into [2^14, 2^15) -- the `wscale` of lemon_split_f16x3 (one host read per weight)."""
    mx = float(w.detach().abs().max())
    if not (mx > 0.0) or mx != mx or mx == float("inf"):
        return 1.0
    import math
    return 2.0 ** (15 - math.frexp(mx)[1])


def split_operand(x, mode="bf16x6", weight=False, wscale=1.0):
    """float32 [..., k] -> the split operand rows of `mode` ([..., 6k] bf16 or [..., 3k] fp16): activation layout, or the weight
    layout (f16x3: of x * wscale)."""
    dtype, seg = _SCHEMES[mode]
    assert x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0
    x = x.contiguous()
    k = x.shape[-1]
    y = torch.empty(x.shape[:-1] + (seg * k,), dtype=dtype, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if mode == "bf16x6":
            _lib.check(lib.lemon_split3_f32(ptr(x), x.numel() // k, k, int(bool(weight)), ptr(y), stream_ptr(x.device)), "lemon_split3_f32")
        else:
            _lib.check(lib.lemon_split_f16x3(ptr(x), x.numel() // k, k, int(bool(weight)), float(wscale), ptr(y), stream_ptr(x.device)),
                       "lemon_split_f16x3")
    return y


def split3(x, weight=False):
    """float32 [..., k] -> bf16 [..., 6k] split operand rows (activation layout, or the weight layout)."""
    return split_operand(x, "bf16x6", weight)


def layer_norm_split(x, weight, bias, eps=1e-5, mode="bf16x6"):
    """LayerNorm whose output is the split activation operand of `mode` (one pass: lemon_layernorm_split3 / _f16x3)."""
    dtype, seg = _SCHEMES[mode]
    assert x.is_cuda and x.dtype == torch.float32
    x = x.contiguous()
    width = x.shape[-1]
    y = torch.empty(x.shape[:-1] + (seg * width,), dtype=dtype, device=x.device)
    lib = _lib.load()
    fn = lib.lemon_layernorm_split3 if mode == "bf16x6" else lib.lemon_layernorm_f16x3
    with torch.cuda.device(x.device):
        _lib.check(fn(ptr(x), ptr(weight.contiguous()), ptr(bias.contiguous()), float(eps), x.numel() // width, width, ptr(y),
                      stream_ptr(x.device)), "lemon_layernorm_" + mode)
    return y


def layer_norm_split3(x, weight, bias, eps=1e-5):
    return layer_norm_split(x, weight, bias, eps, "bf16x6")


def linear_split(xs, ws, bias=None, residual=None, act=None, alpha=1.0):
    """y = act(alpha * xs . ws^T + bias) (+ residual), float32, from split operands of one scheme (the dtype says which):
    bf16 [..., 6k] x [n, 6k] or fp16 [..., 3k] x [n, 3k].  f16x3: the caller folds 1 / wscale into alpha."""
    assert xs.is_cuda and xs.dtype == ws.dtype and xs.dtype in (torch.bfloat16, torch.float16) and xs.shape[-1] == ws.shape[1]
    xs, ws = xs.contiguous(), ws.contiguous()
    ks, n = xs.shape[-1], ws.shape[0]
    m = xs.numel() // ks
    y = torch.empty(xs.shape[:-1] + (n,), dtype=torch.float32, device=xs.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == torch.float32
        residual = residual.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    lib = _lib.load()
    _ensure_linear_tuned(lib, xs.device)
    code = _ACT_CODE[act]
    fn, name = (lib.lemon_linear_bf16x6, "lemon_linear_bf16x6") if xs.dtype == torch.bfloat16 else (lib.lemon_linear_f16x3, "lemon_linear_f16x3")
    with torch.cuda.device(xs.device):
        _lib.check(fn(ptr(xs), ptr(ws), ptr(bias) if bias is not None else None, ptr(residual) if residual is not None else None,
                      m, n, ks, float(alpha), code, ptr(y), stream_ptr(xs.device)), name)
    return y


linear_split3 = linear_split


# ---- the MLP of a block in the hand-written split-fp16 GEMM (gemm_f16x3.hip: tile-major operands, fused fc1 epilogue) ----
def mlp_mode():
    """LEMON_MLP, with LEMON_GEMM=f16x3: 'block' (default): all four GEMMs of a block in the hand-written kernel (gemm_f16x3.hip):
    LayerNorm and attention write its tile-major operands, fc1's epilogue writes fc2's -- with the 16x16x32 kernel of round 4
    this is the fastest mode (35.6 k scores/s against 35.4 k for 'fused' on the same box; in round 3, with the 32x32x16 kernel,
    it was 2 % slower); 'fused': only the MLP there, QKV and the output projection in the library (lemon_linear_f16x3);
    'lib': the library GEMMs throughout, with the separate split pass."""
    import os
    v = os.environ.get("LEMON_MLP", "block").lower()
    if v not in ("block", "fused", "lib"):
        raise ValueError(f"LEMON_MLP={v!r}: expected block, fused or lib")
    return v


def mlp_fused_supported(width, mlp):
    return width % 256 == 0 and mlp % 256 == 0


def block_fused_supported(width, mlp, heads, seq_len):
    """all four GEMMs of a block in the hand-written kernel: additionally 3 * width a multiple of 256 and the HIP attention"""
    return mlp_fused_supported(width, mlp) and (3 * width) % 256 == 0 and width == 64 * heads and seq_len <= ATTENTION_MAX_SEQ


def _tiled_rows(m):
    return (m + 127) // 128 * 128


def pack_weight_t(w, wscale):
    """float32 [n, k] -> the tile-major fp16 weight operand of lemon_linear_f16x3t (of w * wscale)."""
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.shape[0] % 256 == 0 and w.shape[1] % 16 == 0
    w = w.contiguous()
    wt = torch.empty((w.shape[0] * w.shape[1] * 2,), dtype=torch.float16, device=w.device)
    lib = _lib.load()
    with torch.cuda.device(w.device):
        _lib.check(lib.lemon_pack_weight_f16x3t(ptr(w), w.shape[0], w.shape[1], float(wscale), ptr(wt), stream_ptr(w.device)),
                   "lemon_pack_weight_f16x3t")
    return wt


def layer_norm_t(x, weight, bias, eps=1e-5):
    """LayerNorm whose output is the tile-major fp16 activation operand of lemon_linear_f16x3t (flat fp16 tensor)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 16 == 0
    x = x.contiguous()
    width = x.shape[-1]
    m = x.numel() // width
    at = torch.empty((_tiled_rows(m) * width * 2,), dtype=torch.float16, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.lemon_layernorm_f16x3t(ptr(x), ptr(weight.contiguous()), ptr(bias.contiguous()), float(eps), m, width, ptr(at),
                                              stream_ptr(x.device)), "lemon_layernorm_f16x3t")
    return at


def linear_t(at, wt, m, n, k, bias=None, residual=None, act=None, alpha=1.0, out_shape=None):
    """lemon_linear_f16x3t on tile-major operands: act None -> float32 [m, n] (view `out_shape`) = alpha x W^T + bias (+ residual);
    act 'silu' / 'gelu' -> the tile-major activation operand (k' = n) of act(alpha x W^T + bias)."""
    assert at.is_cuda and at.dtype == torch.float16 and wt.dtype == torch.float16
    assert at.numel() == _tiled_rows(m) * k * 2 and wt.numel() == n * k * 2
    lib = _lib.load()
    if bias is not None:
        bias = bias.contiguous()
    if act in ("silu", "gelu"):
        assert residual is None
        out = torch.empty((_tiled_rows(m) * n * 2,), dtype=torch.float16, device=at.device)
    else:
        assert act is None
        out = torch.empty(out_shape if out_shape is not None else (m, n), dtype=torch.float32, device=at.device)
        assert out.numel() == m * n
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.numel() == m * n
            residual = residual.contiguous()
    with torch.cuda.device(at.device):
        _lib.check(lib.lemon_linear_f16x3t(ptr(at), ptr(wt), ptr(bias) if bias is not None else None,
                                           ptr(residual) if residual is not None else None, m, n, k, float(alpha),
                                           _ACT_CODE[act], int(act is not None), ptr(out), stream_ptr(at.device)),
                   "lemon_linear_f16x3t")
    return out


LN_FOLD_MAX_SHIFT = 8.0     # csrc/common.hpp LEMON_LN_FOLD_MAX_SHIFT: rows with |mean| rstd beyond it get a NaN row affine (-> fallback)


_forced_ln_fold = None            # set by ln_fold_forced(): overrides $LEMON_LNFOLD for the calls inside the context


class ln_fold_forced:
    """with ln_fold_forced(False): ... -- the towers run their LayerNorms as kernels for the calls inside, whatever $LEMON_LNFOLD
    says (pipeline.Embedder re-runs a micro-batch whose rows were beyond the fold's mean bound this way first)."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global _forced_ln_fold
        self.prev, _forced_ln_fold = _forced_ln_fold, self.on
        return self

    def __exit__(self, *exc):
        global _forced_ln_fold
        _forced_ln_fold = self.prev
        return False


def ln_fold_enabled():
    """LEMON_LNFOLD (default 1): with LEMON_GEMM=f16x3 and LEMON_MLP=block the LayerNorms in front of QKV and fc1 are folded into
    the hand-written GEMMs (lemon_linear_f16x3t_ln): the output projection / fc2 write the residual stream also as the next
    GEMM's operand together with row statistics, QKV / fc1 apply (rstd, -mean rstd) and the weight-row sums in their epilogue --
    no LayerNorm pass over the token matrix.  0: LayerNorm kernels as before (A/B aid)."""
    import os
    if _forced_ln_fold is not None:
        return _forced_ln_fold
    return os.environ.get("LEMON_LNFOLD", "1") != "0"


def fold_layernorm_weight(w, b, gamma, beta, extra=1.0):
    """The weight side of a folded LayerNorm: LN(x) W^T + b = rstd (x W'^T - mean c) + b' with W' = W diag(gamma).  Returns
    (tile-major packed W' * wscale, 1 / wscale, colsum, bias'): colsum[n] = extra / wscale * sum_k of the PACKED weight row (hi + lo,
    float64 sum), bias' = extra * (b + W beta) -- `extra` is the factor the caller multiplies alpha = 1 / wscale with (QuickGELU's
    1.702 for fc1)."""
    wp = (w.detach() * gamma.detach()[None, :]).contiguous()
    wscale = weight_scale_f16x3(wp)
    wt = pack_weight_t(wp, wscale)
    t = wp * wscale                                       # exact: wscale is a power of two
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)               # split3.hpp split2h<WEIGHT = true>
    colsum = ((hi.double() + lo.double()).sum(1) * (extra / wscale)).float().contiguous()
    bias = (b.detach().double() if b is not None else 0.0) + w.detach().double() @ beta.detach().double()
    return wt, 1.0 / wscale, colsum, (bias * extra).float().contiguous()


def rowstats_t(x, eps):
    """x as the tile-major activation operand + its rows' (rstd, -mean rstd) [m, 2]: the input of a folded LayerNorm for a tensor
    no GEMM epilogue produced (lemon_rowstats_f16x3t)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 16 == 0
    x = x.contiguous()
    width = x.shape[-1]
    m = x.numel() // width
    at = torch.empty((_tiled_rows(m) * width * 2,), dtype=torch.float16, device=x.device)
    aff = torch.empty((m, 2), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.lemon_rowstats_f16x3t(ptr(x), float(eps), m, width, ptr(at), ptr(aff), stream_ptr(x.device)), "lemon_rowstats_f16x3t")
    return at, aff


def ln_finalize(stats, m, width, eps):
    """[m, width / 128, 2] (mean, M2) partials of lemon_linear_f16x3t_ln's emit side -> (rstd, -mean rstd) [m, 2]."""
    assert stats.is_cuda and stats.dtype == torch.float32 and stats.numel() == m * (width // 128) * 2
    aff = torch.empty((m, 2), dtype=torch.float32, device=stats.device)
    lib = _lib.load()
    with torch.cuda.device(stats.device):
        _lib.check(lib.lemon_ln_finalize(ptr(stats), m, width, float(eps), ptr(aff), stream_ptr(stats.device)), "lemon_ln_finalize")
    return aff


def linear_t_ln(at, wt, m, n, k, bias=None, residual=None, act=None, alpha=1.0, out_shape=None, row_aff=None, colsum=None, emit=False):
    """linear_t with a folded LayerNorm on one side (lemon_linear_f16x3t_ln).  row_aff / colsum: this GEMM stands behind a LayerNorm
    whose input `at` holds (see fold_layernorm_weight / rowstats_t / ln_finalize).  emit: returns (out, out as the next GEMM's
    tile-major operand, its row-statistics partials [m, n / 128, 2])."""
    assert at.is_cuda and at.dtype == torch.float16 and wt.dtype == torch.float16
    assert at.numel() == _tiled_rows(m) * k * 2 and wt.numel() == n * k * 2 and k % 32 == 0
    assert (row_aff is None) == (colsum is None) and not (emit and row_aff is not None)
    lib = _lib.load()
    if bias is not None:
        bias = bias.contiguous()
    if act in ("silu", "gelu"):
        assert residual is None and not emit
        out = torch.empty((_tiled_rows(m) * n * 2,), dtype=torch.float16, device=at.device)
    else:
        assert act is None
        out = torch.empty(out_shape if out_shape is not None else (m, n), dtype=torch.float32, device=at.device)
        assert out.numel() == m * n
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.numel() == m * n
            residual = residual.contiguous()
    if row_aff is not None:
        assert row_aff.dtype == torch.float32 and row_aff.numel() == 2 * m and colsum.dtype == torch.float32 and colsum.numel() == n
        row_aff, colsum = row_aff.contiguous(), colsum.contiguous()
    et = st = None
    if emit:
        et = torch.empty((_tiled_rows(m) * n * 2,), dtype=torch.float16, device=at.device)
        st = torch.empty((m, n // 128, 2), dtype=torch.float32, device=at.device)
    with torch.cuda.device(at.device):
        _lib.check(lib.lemon_linear_f16x3t_ln(ptr(at), ptr(wt), ptr(bias) if bias is not None else None,
                                              ptr(residual) if residual is not None else None, m, n, k, float(alpha),
                                              _ACT_CODE[act], int(act is not None), ptr(out),
                                              ptr(row_aff) if row_aff is not None else None, ptr(colsum) if colsum is not None else None,
                                              ptr(et) if emit else None, ptr(st) if emit else None, stream_ptr(at.device)),
                   "lemon_linear_f16x3t_ln")
    return (out, et, st) if emit else out


def chain_operand_residual():
    """LEMON_CHAIN_RES: how the residual stream travels inside a block chain (ln_fold_enabled).
    2 (default): as the GEMMs' tile-major operand only -- the output projection and fc2 leave their result as the next GEMM's
       operand + row statistics (no fp32 tensor: 6 instead of 10 bytes written per element) and take their residual from the
       operand the GEMM in front left (hi + lo 2^-11: 22 significant bits per hop, the size of the terms the split products drop
       anyway); only the last chained block also writes fp32 (the pooled last block and the final LayerNorm read it);
    1: only the stream between the output projection and fc2 is carried that way (fc2 writes fp32 at every block boundary);
    0: fp32 residual stream between all GEMMs (A/B aid)."""
    import os
    v = os.environ.get("LEMON_CHAIN_RES", "2")
    return 0 if v == "0" else (1 if v == "1" else 2)


def linear_t_chain(at, wt, m, n, k, bias=None, residual=None, residual_t=None, alpha=1.0, out_shape=None, fp32_out=True):
    """lemon_linear_f16x3t_chain: alpha x W^T + bias + residual -> (fp32 [m, n] or None, the same as the next GEMM's tile-major
    operand, its row-statistics partials [m, n / 128, 2]); the residual is fp32 [m, n] (`residual`) or a tile-major operand
    (`residual_t`)."""
    assert at.is_cuda and at.dtype == torch.float16 and wt.dtype == torch.float16
    assert at.numel() == _tiled_rows(m) * k * 2 and wt.numel() == n * k * 2 and k % 32 == 0
    assert residual is None or residual_t is None
    lib = _lib.load()
    if bias is not None:
        bias = bias.contiguous()
    out = None
    if fp32_out:
        out = torch.empty(out_shape if out_shape is not None else (m, n), dtype=torch.float32, device=at.device)
        assert out.numel() == m * n
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.numel() == m * n
        residual = residual.contiguous()
    if residual_t is not None:
        assert residual_t.dtype == torch.float16 and residual_t.numel() == _tiled_rows(m) * n * 2
    et = torch.empty((_tiled_rows(m) * n * 2,), dtype=torch.float16, device=at.device)
    st = torch.empty((m, n // 128, 2), dtype=torch.float32, device=at.device)
    with torch.cuda.device(at.device):
        _lib.check(lib.lemon_linear_f16x3t_chain(ptr(at), ptr(wt), ptr(bias) if bias is not None else None,
                                                 ptr(residual) if residual is not None else None,
                                                 ptr(residual_t) if residual_t is not None else None, m, n, k, float(alpha),
                                                 ptr(out) if out is not None else None, ptr(et), ptr(st), stream_ptr(at.device)),
                   "lemon_linear_f16x3t_chain")
    return out, et, st


def gemm_profiling(on):
    """HIP events around every lemon_linear_f16x3t launch from now on (bench.py: the roofline of the step's dominant kernel)."""
    _lib.check(_lib.load().lemon_linear_f16x3t_set_profiling(int(bool(on))), "lemon_linear_f16x3t_set_profiling")


def gemm_profile_read():
    """{launches, kernel_ms, flops} of the bracketed launches since the last read (waits for them; rewinds the event pool)."""
    n, ms, fl = ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    _lib.check(_lib.load().lemon_linear_f16x3t_profile_read(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)),
               "lemon_linear_f16x3t_profile_read")
    return {"launches": int(n.value), "kernel_ms": float(ms.value), "flops": float(fl.value)}


def attention_t(qkv, heads, causal=False):
    """attention() whose output is the tile-major fp16 activation operand of lemon_linear_f16x3t (rows = B*L, k = heads*64)."""
    assert qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 3
    qkv = qkv.contiguous()
    B, L, W3 = qkv.shape
    assert W3 == 3 * heads * 64 and L <= ATTENTION_MAX_SEQ
    out = torch.empty((_tiled_rows(B * L) * heads * 64 * 2,), dtype=torch.float16, device=qkv.device)
    lib = _lib.load()
    with torch.cuda.device(qkv.device):
        _lib.check(lib.lemon_attention_f16x3t(ptr(qkv), B, L, heads, 64, int(bool(causal)), ptr(out), stream_ptr(qkv.device)),
                   "lemon_attention_f16x3t")
    return out


def unpack_act_t(at, m, k):
    """tile-major activation operand -> float32 [m, k] (hi + lo 2^-11)."""
    y = torch.empty((m, k), dtype=torch.float32, device=at.device)
    lib = _lib.load()
    with torch.cuda.device(at.device):
        _lib.check(lib.lemon_unpack_act_f16x3t(ptr(at), m, k, ptr(y), stream_ptr(at.device)), "lemon_unpack_act_f16x3t")
    return y


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dimension of a float32 CUDA tensor in one HIP pass (lemon_layernorm_f32)."""
    assert x.is_cuda and x.dtype == torch.float32
    x = x.contiguous()
    width = x.shape[-1]
    y = torch.empty_like(x)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.lemon_layernorm_f32(ptr(x), ptr(weight.contiguous()), ptr(bias.contiguous()), float(eps),
                                           x.numel() // width, width, ptr(y), stream_ptr(x.device)), "lemon_layernorm_f32")
    return y


def vision_tokens_ln(patches, cls, pos, ln_weight, ln_bias, eps=1e-5):
    """[B, nP, W] patch embeddings -> pre-LayerNormed token matrix [B, nP+1, W] (class token + positions) in one pass;
    ln_weight = ln_bias = None: token assembly without the LayerNorm (timm ViT)."""
    assert patches.is_cuda and patches.dtype == torch.float32 and patches.dim() == 3
    patches = patches.contiguous()
    B, nP, W = patches.shape
    y = torch.empty((B, nP + 1, W), dtype=torch.float32, device=patches.device)
    lib = _lib.load()
    with torch.cuda.device(patches.device):
        _lib.check(lib.lemon_vision_tokens_ln(ptr(patches), ptr(cls.contiguous()), ptr(pos.contiguous()),
                                              ptr(ln_weight.contiguous()) if ln_weight is not None else None,
                                              ptr(ln_bias.contiguous()) if ln_bias is not None else None, float(eps), B, nP + 1, W, ptr(y),
                                              stream_ptr(patches.device)), "lemon_vision_tokens_ln")
    return y


def text_tokens(input_ids, seq_len, tok_weight, pos):
    """tok_weight[ids[:, :seq_len]] + pos[:seq_len] -> [B, seq_len, W] in one pass (ids int64 CUDA [B, ctx])."""
    assert input_ids.is_cuda and input_ids.dtype == torch.int64 and input_ids.dim() == 2 and input_ids.stride(1) == 1
    B, W = input_ids.shape[0], tok_weight.shape[1]
    y = torch.empty((B, seq_len, W), dtype=torch.float32, device=input_ids.device)
    lib = _lib.load()
    with torch.cuda.device(input_ids.device):
        _lib.check(lib.lemon_text_tokens(ptr(input_ids), input_ids.stride(0), ptr(tok_weight.contiguous()), ptr(pos.contiguous()), B,
                                         int(seq_len), W, tok_weight.shape[0], ptr(y), stream_ptr(input_ids.device)),
                   "lemon_text_tokens")
    return y


def linear_dump_tuned(path):
    """Write the solution choices made so far in this process (tools/tune_gemms.py)."""
    return _lib.load().lemon_linear_dump_tuned(str(path).encode())


def attention(qkv, heads, causal=False):
    """Fused self-attention on the packed projection output qkv [B, L, 3*W] (float32, contiguous,
    W = heads*64) -> [B, L, W]; the HIP kernel behind LemonCLIP's blocks."""
    assert qkv.is_cuda and qkv.dtype == torch.float32 and qkv.is_contiguous() and qkv.dim() == 3
    B, L, W3 = qkv.shape
    W = W3 // 3
    assert W3 == 3 * W and W == heads * 64 and L <= ATTENTION_MAX_SEQ
    out = torch.empty((B, L, W), dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    with torch.cuda.device(qkv.device):
        _lib.check(lib.lemon_attention_f32(ptr(qkv), B, L, heads, 64, int(bool(causal)), ptr(out),
                                           stream_ptr(qkv.device)), "lemon_attention_f32")
    return out


def attention_split(qkv, heads, causal=False, mode="bf16x6"):
    """attention() whose output is the split activation operand of `mode`: [B, L, 6*W] bf16 (lemon_attention_split3) or
    [B, L, 3*W] fp16 (lemon_attention_f16x3)."""
    dtype, seg = _SCHEMES[mode]
    assert qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 3
    qkv = qkv.contiguous()
    B, L, W3 = qkv.shape
    assert W3 == 3 * heads * 64 and L <= ATTENTION_MAX_SEQ
    out = torch.empty((B, L, seg * heads * 64), dtype=dtype, device=qkv.device)
    lib = _lib.load()
    fn = lib.lemon_attention_split3 if mode == "bf16x6" else lib.lemon_attention_f16x3
    with torch.cuda.device(qkv.device):
        _lib.check(fn(ptr(qkv), B, L, heads, 64, int(bool(causal)), ptr(out), stream_ptr(qkv.device)), "lemon_attention_" + mode)
    return out


def attention_split3(qkv, heads, causal=False):
    return attention_split(qkv, heads, causal, "bf16x6")



def paired_distance(metric, a, b):
    a, b = dev_f32(a, "a"), dev_f32(b, "b")
    assert a.shape == b.shape and a.dim() == 2
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    lib = _lib.load()
    with torch.cuda.device(a.device):
        _lib.check(lib.lemon_paired_distance(metric_id(metric), ptr(a), ptr(b), a.shape[0], a.shape[1],
                                             ptr(out), stream_ptr(a.device)), "lemon_paired_distance")
    return out


def d1_normalized(metric, q_img, cls_txt, noisy_label):
    q, c = dev_f32(q_img, "q_img"), dev_f32(cls_txt, "cls_txt")
    lab = noisy_label.to(device=q.device, dtype=torch.int32).contiguous()
    out = torch.empty(q.shape[0], dtype=torch.float32, device=q.device)
    lib = _lib.load()
    with torch.cuda.device(q.device):
        _lib.check(lib.lemon_d1_normalized(metric_id(metric), ptr(q), q.shape[0], q.shape[1], ptr(c),
                                           c.shape[0], ptr(lab), ptr(out), stream_ptr(q.device)),
                   "lemon_d1_normalized")
    return out


HP_ORDER = ("beta", "gamma", "tau_1_n", "tau_2_n", "tau_1_m", "tau_2_m")


def lemon_score(rec, hparams, return_dn=False):
    """Score aggregation on device arrays.  rec: dict with d_1 [n], D_n, dists_tr_n, dists_n, D_m,
    dists_tr_m, dists_m [n,k] CUDA tensors.  Returns float64 CUDA tensors."""
    names = ("d_1", "D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")
    arrs = [dev_f32(rec[nm], nm) for nm in names]
    n, k = arrs[1].shape
    dev = arrs[0].device
    score = torch.empty(n, dtype=torch.float64, device=dev)
    dn = torch.empty(n, dtype=torch.float64, device=dev)
    dm = torch.empty(n, dtype=torch.float64, device=dev)
    hp = (ctypes.c_double * 6)(*[float(hparams[h]) for h in HP_ORDER])
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.lemon_score(*[ptr(a) for a in arrs], n, k, hp, ptr(score), ptr(dn), ptr(dm),
                                   stream_ptr(dev)), "lemon_score")
    return (score, dn, dm) if return_dn else score


def _stack_col(df, col, device):
    return torch.from_numpy(np.stack(df[col].values).astype(np.float32, copy=False)).to(device)


def calc_scores_given_hparams_vectorized(df, best_hparams, return_dn=False, torch_arr=False, device="cuda"):
    """Drop-in for lib/metrics/utils.py:47-82 on the reference's DataFrame schema
    (run_lemon.py:291-307): returns numpy arrays (torch CPU tensors when torch_arr)."""
    rec = {c: _stack_col(df, c, device) for c in ("D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")}
    rec["d_1"] = torch.from_numpy(np.asarray(df["d_1"].values, dtype=np.float32)).to(device)
    s, dn, dm = lemon_score(rec, best_hparams, return_dn=True)
    # d_1 is float64 in the reference frame (`d1.item()`): add it back at full precision
    s = s - rec["d_1"].double() + torch.from_numpy(np.asarray(df["d_1"].values, dtype=np.float64)).to(device)
    conv = (lambda t: t.cpu()) if torch_arr else (lambda t: t.cpu().numpy())
    if return_dn:
        return conv(s), conv(dn), conv(dm)
    return conv(s)


calc_scores_given_hparams = calc_scores_given_hparams_vectorized  # loop twin, lib/metrics/utils.py:21-45
