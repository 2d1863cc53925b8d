"""Flat exact-search index with the faiss surface run_lemon.py uses.

  faiss.IndexFlatIP(d) / faiss.IndexFlatL2(d)     run_lemon.py:167-168,171-172
  index.add(x)                                    run_lemon.py:175-176
  index.search(x, k) -> (D, I)                    run_lemon.py:235-236
  index.ntotal, index.d

numpy in -> numpy out (like faiss; one H2D/D2H per call), torch CUDA tensor in -> CUDA
tensors out (embeddings stay in HBM, which is the point of this rebuild).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .ops import dev_f32, ptr, stream_ptr


class _IndexFlat:
    metric = None

    def __init__(self, d, device=None):
        if int(d) <= 0:
            raise ValueError("d must be positive")
        self.d = int(d)
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.LemonHipError("no HIP device: lemon_amd.IndexFlat* runs only on the GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.lemon_index_create(self.metric, self.d, ctypes.byref(h)), "lemon_index_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.lemon_index_free(h)
            except Exception:
                pass
            self._h = None

    @property
    def ntotal(self):
        return int(self._lib.lemon_index_ntotal(self._h))

    def set_algo(self, algo):
        _lib.check(self._lib.lemon_index_set_algo(self._h, int(algo)), "lemon_index_set_algo")

    def set_query_dedup(self, enabled=True):
        """Fold identical query rows into one search each (exact; on by default -- see include/lemon_hip.h)."""
        _lib.check(self._lib.lemon_index_set_query_dedup(self._h, int(bool(enabled))), "lemon_index_set_query_dedup")

    def last_search_info(self):
        info = _lib.SearchInfo()
        _lib.check(self._lib.lemon_index_last_search_info(self._h, ctypes.byref(info)), "last_search_info")
        return {f: getattr(info, f) for f, _ in info._fields_}

    def set_profiling(self, enabled=True):
        _lib.check(self._lib.lemon_index_set_profiling(self._h, int(bool(enabled))), "lemon_index_set_profiling")

    def profile_read(self):
        """(launches, kernel_ms, algo_flops, algo_bytes) of the scan kernel since the last read."""
        n = ctypes.c_int64()
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.check(self._lib.lemon_index_profile_read(self._h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl),
                                                      ctypes.byref(by)), "lemon_index_profile_read")
        return dict(launches=n.value, kernel_ms=ms.value, algo_flops=fl.value, algo_bytes=by.value)

    def _to_dev(self, x, what):
        if isinstance(x, np.ndarray):
            # faiss asserts a C-contiguous float32 [n, d] array
            assert x.dtype == np.float32, f"{what}: float32 required (faiss contract), got {x.dtype}"
            assert x.ndim == 2 and x.shape[1] == self.d, f"{what}: expected shape [n, {self.d}], got {x.shape}"
            return torch.from_numpy(np.ascontiguousarray(x)).to(self.device), True
        t = dev_f32(x, what)
        assert t.dim() == 2 and t.shape[1] == self.d, f"{what}: expected shape [n, {self.d}], got {tuple(t.shape)}"
        assert t.device == self.device, f"{what}: tensor on {t.device}, index on {self.device}"
        return t, False

    def add(self, x):
        t, _ = self._to_dev(x, "add(x)")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.lemon_index_add(self._h, ptr(t), t.shape[0], stream_ptr(self.device)),
                       "lemon_index_add")

    def search(self, x, k):
        k = int(k)  # run_lemon.py passes k + (sname == 'train'), an int + bool
        if not 1 <= k <= _lib.MAX_K_DEEP:
            raise ValueError(f"k must be in [1, {_lib.MAX_K_DEEP}], got {k}")
        t, was_numpy = self._to_dev(x, "search(x)")
        nq = t.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.lemon_index_search(self._h, ptr(t), nq, k, ptr(D), ptr(I),
                                                    stream_ptr(self.device)), "lemon_index_search")
        if was_numpy:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    def data(self):
        """Zero-copy CUDA view [ntotal, d] of the stored rows (valid until the next add / until the index dies)."""
        n = self.ntotal
        if n == 0:
            return torch.empty((0, self.d), dtype=torch.float32, device=self.device)

        class _View:       # __cuda_array_interface__ v2: torch wraps the device pointer without copying
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (n, self.d), "typestr": "<f4", "data": (int(self._lib.lemon_index_data(self._h)), False),
                                      "version": 2, "strides": None}
        with torch.cuda.device(self.device):
            return torch.as_tensor(v, device=self.device)

    def reconstruct_n(self, i0=0, n=None):
        """Rows [i0, i0+n) as stored (faiss `reconstruct_n`): a CUDA float32 copy [n, d]."""
        n = self.ntotal - i0 if n is None else int(n)
        assert 0 <= i0 and i0 + n <= self.ntotal
        return self.data()[i0:i0 + n].clone()


class IndexFlatIP(_IndexFlat):
    metric = _lib.METRIC_IP


class IndexFlatL2(_IndexFlat):
    metric = _lib.METRIC_L2
