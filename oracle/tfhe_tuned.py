"""ctypes wrapper of oracle/libtfhe_tuned.so: the TUNED CPU baseline (AVX-512 IFMA, eight bootstraps per vector, OpenMP) that
bench.py prints beside the scalar oracle's figure.  Bench/test infrastructure -- never the checker, never the product: it
is itself held to the oracle word for word (tests/test_oracle_tfhe.py)."""
import ctypes as C
import os

import numpy as np

from . import tfhe_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        orc.build()
        L = C.CDLL(os.path.join(_HERE, "libtfhe_tuned.so"))
        L.tuned_supported.restype = C.c_int
        L.tuned_create.restype = C.c_void_p
        L.tuned_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tuned_destroy.argtypes = [C.c_void_p]
        L.tuned_bootstrap_batch.restype = C.c_int
        L.tuned_bootstrap_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def supported():
    """Does this CPU have AVX-512 IFMA?"""
    return bool(lib().tuned_supported())


class Tuned:
    """Keyed from an `oracle.tfhe_oracle.Oracle` (same keys, same conventions, so the same ciphertexts)."""

    def __init__(self, oracle):
        self.o = oracle
        keys = oracle.keys()
        self._bsk = np.ascontiguousarray(keys["bsk"], np.uint64)
        self._ksk = np.ascontiguousarray(keys["ksk"], np.uint64)          # borrowed by the C side: keep alive
        cp = orc._P(bsk_group=oracle.group, reserved=0, **oracle.p)
        self._h = lib().tuned_create(C.byref(cp), self._bsk.ctypes.data, self._ksk.ctypes.data)
        if not self._h:
            raise ValueError("the tuned baseline does not cover this CPU / parameter set (AVX-512 IFMA, k = 1, one key bit per step)")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.tuned_destroy(self._h)
            self._h = None

    def bootstrap_batch(self, cts, tables, table_ids=None, threads=0):
        """-> (output ciphertexts, threads used); the oracle's `bootstrap_batch` contract."""
        cts = np.ascontiguousarray(cts, np.uint64)
        count = cts.shape[0]
        tvs = np.empty((len(tables), self.o.N), np.uint64)
        post = np.empty(len(tables), np.uint64)
        for i, t in enumerate(tables):
            tvs[i], post[i] = self.o.build_tv(t)
        ids = None if table_ids is None else np.ascontiguousarray(table_ids, np.uint32)
        out = np.empty_like(cts)
        used = lib().tuned_bootstrap_batch(self._h, cts.ctypes.data, None if ids is None else ids.ctypes.data, tvs.ctypes.data,
                                           post.ctypes.data, count, out.ctypes.data, int(threads))
        return out, used
