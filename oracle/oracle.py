"""ctypes front-end of the CPU oracle (oracle/lemon_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from lemon_amd/.  See the header of
lemon_oracle.c for the numeric contract and the parity status ("parity unpinned"
at the faiss boundary, pinned by reference-generated golden vectors elsewhere).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LEMON_ORACLE_SO: another build of lemon_oracle.c to load instead (tests/test_sanitizers.py: the ASan + UBSan build)
_SO = os.environ.get("LEMON_ORACLE_SO") or os.path.join(_HERE, "liblemon_oracle.so")

METRIC_IP = 0
METRIC_L2 = 1
_METRICS = {"cosine": METRIC_IP, "ip": METRIC_IP, "euclidean": METRIC_L2, "l2": METRIC_L2,
            METRIC_IP: METRIC_IP, METRIC_L2: METRIC_L2}


def build(force=False):
    if os.environ.get("LEMON_ORACLE_SO"):
        return _SO
    src = os.path.join(_HERE, "lemon_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblemon_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def usable_cores():
    """CPU cores this process may actually use: min(affinity mask, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def set_threads(n):
    lib().lo_set_threads(ctypes.c_int(int(n)))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def normalize_rows(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().lo_normalize_rows(_p(x), ctypes.c_int64(x.shape[0]), ctypes.c_int(x.shape[1]), _p(y))
    return y


def paired_distance(metric, a, b):
    a, b = _f32(a), _f32(b)
    assert a.shape == b.shape
    out = np.empty(a.shape[0], dtype=np.float32)
    lib().lo_paired_distance(ctypes.c_int(_METRICS[metric]), _p(a), _p(b),
                             ctypes.c_int64(a.shape[0]), ctypes.c_int(a.shape[1]), _p(out))
    return out


def dot_chain(a, b):
    a, b = _f32(a).ravel(), _f32(b).ravel()
    f = lib().lo_dot_chain
    f.restype = ctypes.c_float
    return np.float32(f(_p(a), _p(b), ctypes.c_int(a.size)))


def knn(metric, X, Q, k):
    """Exact flat search: returns (D [nq,k] f32, I [nq,k] i64), best first."""
    X, Q = _f32(X), _f32(Q)
    n, d = X.shape if X.ndim == 2 else (0, Q.shape[1])
    nq = Q.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().lo_knn(ctypes.c_int(_METRICS[metric]), _p(X), ctypes.c_int64(n), ctypes.c_int(d),
                 _p(Q), ctypes.c_int64(nq), ctypes.c_int(k), _p(D), _p(I))
    return D, I


def neighbors(metric, img_tr, txt_tr, q_img, q_txt, k, drop_self=False, in_db=None,
              discrete=False, tr_label_id=None, q_label_id=None, dists_tr=None):
    """Restatement of run_lemon.py:238-307 for one split.  Returns a dict of arrays."""
    img_tr, txt_tr, q_img, q_txt = _f32(img_tr), _f32(txt_tr), _f32(q_img), _f32(q_txt)
    n_tr, d = img_tr.shape
    nq = q_img.shape[0]
    m = _METRICS[metric]
    if dists_tr is None:
        dists_tr = paired_distance(m, txt_tr, img_tr)
    dists_tr = _f32(dists_tr)
    in_db = np.ascontiguousarray(in_db if in_db is not None else np.ones(nq), dtype=np.uint8)
    trl = np.ascontiguousarray(tr_label_id if tr_label_id is not None else np.zeros(n_tr), dtype=np.int32)
    ql = np.ascontiguousarray(q_label_id if q_label_id is not None else np.zeros(nq), dtype=np.int32)
    out = {"d_1": np.empty(nq, np.float32)}
    for nm in ("D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"):
        out[nm] = np.empty((nq, k), np.float32)
    out["I_n"] = np.empty((nq, k), np.int64)
    out["I_m"] = np.empty((nq, k), np.int64)
    lib().lo_neighbors(ctypes.c_int(m), _p(img_tr), _p(txt_tr), _p(dists_tr), ctypes.c_int64(n_tr),
                       ctypes.c_int(d), _p(q_img), _p(q_txt), ctypes.c_int64(nq), ctypes.c_int(k),
                       ctypes.c_int(int(drop_self)), _p(in_db), ctypes.c_int(int(discrete)),
                       _p(trl), _p(ql), _p(out["d_1"]), _p(out["D_n"]), _p(out["dists_n"]),
                       _p(out["dists_tr_n"]), _p(out["I_n"]), _p(out["D_m"]), _p(out["dists_m"]),
                       _p(out["dists_tr_m"]), _p(out["I_m"]))
    out["dists_tr"] = dists_tr
    return out


def d1_normalized(metric, q_img, cls_txt, noisy_label):
    q_img, cls_txt = _f32(q_img), _f32(cls_txt)
    lab = np.ascontiguousarray(noisy_label, dtype=np.int32)
    out = np.empty(q_img.shape[0], np.float32)
    lib().lo_d1_normalized(ctypes.c_int(_METRICS[metric]), _p(q_img), ctypes.c_int64(q_img.shape[0]),
                           ctypes.c_int(q_img.shape[1]), _p(cls_txt), ctypes.c_int(cls_txt.shape[0]),
                           _p(lab), _p(out))
    return out


def discrepancy(method, E_tr, txt_tr, qv, q_txt, k, is_train=False):
    """lib/baselines/discrepancy_baseline.py:164-242.  method: 'dis' | 'div'."""
    E_tr, txt_tr, qv, q_txt = _f32(E_tr), _f32(txt_tr), _f32(qv), _f32(q_txt)
    out = np.empty(q_txt.shape[0], np.float32)
    lib().lo_discrepancy(ctypes.c_int({"dis": 0, "div": 1}[method]), _p(E_tr), _p(txt_tr),
                         ctypes.c_int64(E_tr.shape[0]), ctypes.c_int(E_tr.shape[1]), _p(qv), _p(q_txt),
                         ctypes.c_int64(q_txt.shape[0]), ctypes.c_int(k), ctypes.c_int(int(is_train)), _p(out))
    return out


HP_ORDER = ("beta", "gamma", "tau_1_n", "tau_2_n", "tau_1_m", "tau_2_m")


def score(rec, hparams, return_dn=False):
    """lib/metrics/utils.py:47-82 on a dict of per-sample arrays (as returned by neighbors())."""
    hp = np.array([float(hparams[h]) for h in HP_ORDER], dtype=np.float64)
    n, k = rec["D_n"].shape
    s = np.empty(n, np.float64)
    dn = np.empty(n, np.float64)
    dm = np.empty(n, np.float64)
    arrs = [_f32(rec[nm]) for nm in ("d_1", "D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")]
    lib().lo_score(*[_p(a) for a in arrs], ctypes.c_int64(n), ctypes.c_int(k), _p(hp), _p(s), _p(dn), _p(dm))
    return (s, dn, dm) if return_dn else s


def auroc(y, score_):
    """sklearn.metrics.roc_auc_score (lib/metrics/utils.py:408-412) via the rank statistic
    with average ranks for ties (what sklearn's trapezoid on the ROC curve equals)."""
    from scipy.stats import rankdata
    y = np.asarray(y).astype(bool)
    r = rankdata(np.asarray(score_, dtype=np.float64))
    n1 = int(y.sum())
    n0 = y.size - n1
    return float((r[y].sum() - n1 * (n1 + 1) / 2.0) / (n0 * n1))
