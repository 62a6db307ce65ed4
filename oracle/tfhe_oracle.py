"""ctypes wrapper of the C oracle (oracle/tfhe_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never
by tfhe_fbs_map_amd.  Ciphertext-level parity is "unpinned" against third parties
(see tfhe_oracle.h); decrypted results are pinned to the reference's goldens."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtfhe_oracle.so")
Q = 0x3FFFFFF84001          # 2^46 - 62*2^13 + 1
QBITS = 46


def build(force=False):
    src = os.path.join(_HERE, "tfhe_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return LIB_PATH


class _P(C.Structure):
    _fields_ = [(f, C.c_uint32) for f in
                ("n", "log_n_poly", "k", "l_bsk", "beta_bsk", "t_ksk", "gamma_ksk", "p_msg")] + \
               [("sigma_lwe", C.c_uint64), ("sigma_glwe", C.c_uint64), ("bsk_group", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, u64, u32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
        L.orc_gl_mul.restype = u64
        L.orc_gl_mul.argtypes = [u64, u64]
        L.orc_gl_mul_slow.restype = u64
        L.orc_gl_mul_slow.argtypes = [u64, u64]
        L.orc_gl_pow.restype = u64
        L.orc_gl_pow.argtypes = [u64, u64]
        L.orc_negacyclic_mul_schoolbook.argtypes = [vp, vp, vp, u32]
        L.orc_negacyclic_mul_ntt.argtypes = [vp, vp, vp, u32]
        L.orc_rand64.restype = u64
        L.orc_rand64.argtypes = [u64, u64, u64]
        L.orc_noise.restype = C.c_int64
        L.orc_noise.argtypes = [u64, u64, u64, u64]
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(_P), u64]
        L.orc_destroy.argtypes = [vp]
        L.orc_keygen.argtypes = [vp]
        L.orc_set_keys.argtypes = [vp, vp, vp, vp, vp]
        for f in ("orc_sk_lwe", "orc_sk_glwe", "orc_bsk", "orc_ksk"):
            getattr(L, f).restype = C.POINTER(u64)
            getattr(L, f).argtypes = [vp]
        L.orc_delta_half.restype = u64
        L.orc_delta_half.argtypes = [vp]
        L.orc_encrypt.argtypes = [vp, vp, sz, u64, vp]
        L.orc_decrypt.argtypes = [vp, vp, sz, vp]
        L.orc_phase.argtypes = [vp, vp, sz, vp]
        L.orc_build_tv.restype = C.c_int
        L.orc_build_tv.argtypes = [vp, vp, u32, vp, C.POINTER(u64)]
        L.orc_lincomb.argtypes = [vp, vp, vp, u32, C.c_int64, vp]
        L.orc_keyswitch.argtypes = [vp, vp, vp]
        L.orc_modswitch.argtypes = [vp, vp, vp]
        L.orc_blind_rotate.argtypes = [vp, vp, vp, vp]
        L.orc_sample_extract.argtypes = [vp, vp, u64, vp]
        L.orc_bootstrap.argtypes = [vp, vp, vp, u64, vp]
        L.orc_tv0.argtypes = [vp, vp]
        L.orc_build_tv_diff.restype = C.c_int
        L.orc_build_tv_diff.argtypes = [vp, vp, u32, vp, C.POINTER(u64)]
        L.orc_multi_extract.argtypes = [vp, vp, vp, u64, vp]
        L.orc_bootstrap_batch.restype = C.c_int
        L.orc_bootstrap_batch.argtypes = [vp, vp, vp, vp, vp, sz, vp, C.c_int]
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Oracle:
    """Same parameter fields as tfhe_fbs_map_amd.Params (passed as a dict or any object
    with those attributes)."""

    FIELDS = ("n", "log_n_poly", "k", "l_bsk", "beta_bsk", "t_ksk", "gamma_ksk", "p_msg", "sigma_lwe", "sigma_glwe")

    def __init__(self, params, seed=1, keygen=True):
        get = (lambda f: params[f]) if isinstance(params, dict) else (lambda f: getattr(params, f))
        self.p = {f: int(get(f)) for f in self.FIELDS}
        try:
            self.group = int(get("bsk_group"))
        except (KeyError, AttributeError):
            self.group = 1
        self.N = 1 << self.p["log_n_poly"]
        self.D = self.p["k"] * self.N
        self.ctw = self.D + 1
        cp = _P(bsk_group=self.group, reserved=0, **self.p)
        self._h = lib().orc_create(C.byref(cp), seed)
        if not self._h:
            raise ValueError("oracle rejected the parameter set")
        if keygen:
            lib().orc_keygen(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_destroy(self._h)
            self._h = None

    def key_sizes(self):
        p = self.p
        ggsw = p["n"] // 2 * 3 if self.group == 2 else p["n"]
        return (p["n"], self.D, ggsw * (p["k"] + 1) * p["l_bsk"] * (p["k"] + 1) * self.N,
                self.D * p["t_ksk"] * (p["n"] + 1))

    def keys(self):
        L = lib()
        out = {}
        for name, fn, sz in zip(("sk_lwe", "sk_glwe", "bsk", "ksk"),
                                (L.orc_sk_lwe, L.orc_sk_glwe, L.orc_bsk, L.orc_ksk), self.key_sizes()):
            out[name] = np.ctypeslib.as_array(fn(self._h), shape=(sz,)).copy()
        return out

    def set_keys(self, sk_lwe, sk_glwe, bsk, ksk):
        a = [_c(x, np.uint64) for x in (sk_lwe, sk_glwe, bsk, ksk)]
        lib().orc_set_keys(self._h, *[x.ctypes.data for x in a])

    @property
    def delta_half(self):
        return lib().orc_delta_half(self._h)

    def encrypt(self, msgs, nonce0=0):
        msgs = _c(msgs, np.int64)
        cts = np.empty(msgs.shape + (self.ctw,), np.uint64)
        lib().orc_encrypt(self._h, msgs.ctypes.data, msgs.size, nonce0, cts.ctypes.data)
        return cts

    def decrypt(self, cts):
        cts = _c(cts, np.uint64)
        out = np.empty(cts.shape[:-1], np.int64)
        lib().orc_decrypt(self._h, cts.ctypes.data, out.size, out.ctypes.data)
        return out

    def phase(self, cts):
        cts = _c(cts, np.uint64)
        out = np.empty(cts.shape[:-1], np.uint64)
        lib().orc_phase(self._h, cts.ctypes.data, out.size, out.ctypes.data)
        return out

    def build_tv(self, table):
        t = _c(table, np.int32)
        tv = np.empty(self.N, np.uint64)
        post = C.c_uint64()
        rc = lib().orc_build_tv(self._h, t.ctypes.data, len(t), tv.ctypes.data, C.byref(post))
        if rc != 0:
            raise ValueError(f"table {list(t)} is not negacyclic-valid for p={self.p['p_msg']}")
        return tv, post.value

    def lincomb(self, srcs, coefs, const_coef=0):
        srcs = [_c(s, np.uint64) for s in srcs]
        ptrs = (C.c_void_p * len(srcs))(*[s.ctypes.data for s in srcs])
        coefs = _c(coefs, np.int64)
        out = np.empty(self.ctw, np.uint64)
        lib().orc_lincomb(self._h, ptrs, coefs.ctypes.data, len(srcs), const_coef, out.ctypes.data)
        return out

    def keyswitch(self, ct):
        ct = _c(ct, np.uint64)
        out = np.empty(self.p["n"] + 1, np.uint64)
        lib().orc_keyswitch(self._h, ct.ctypes.data, out.ctypes.data)
        return out

    def modswitch(self, small):
        small = _c(small, np.uint64)
        out = np.empty(self.p["n"] + 1, np.uint32)
        lib().orc_modswitch(self._h, small.ctypes.data, out.ctypes.data)
        return out

    def blind_rotate(self, ms, tv):
        ms, tv = _c(ms, np.uint32), _c(tv, np.uint64)
        acc = np.empty((self.p["k"] + 1) * self.N, np.uint64)
        lib().orc_blind_rotate(self._h, ms.ctypes.data, tv.ctypes.data, acc.ctypes.data)
        return acc

    # ---- several tables on one blind rotation (multi-value bootstrap, see tfhe_oracle.c) ----
    def tv0(self):
        tv = np.empty(self.N, np.uint64)
        lib().orc_tv0(self._h, tv.ctypes.data)
        return tv

    def build_tv_diff(self, table):
        """(D_F as int32[N], post_add) with TV_F = TV_0 * D_F."""
        t = _c(table, np.int32)
        diff = np.empty(self.N, np.int32)
        post = C.c_uint64()
        rc = lib().orc_build_tv_diff(self._h, t.ctypes.data, len(t), diff.ctypes.data, C.byref(post))
        if rc != 0:
            raise ValueError(f"table {list(t)} is not negacyclic-valid for p={self.p['p_msg']}")
        return diff, post.value

    def multi_extract(self, acc, diff, post_add=0):
        acc, diff = _c(acc, np.uint64), _c(diff, np.int32)
        out = np.empty(self.ctw, np.uint64)
        lib().orc_multi_extract(self._h, acc.ctypes.data, diff.ctypes.data, post_add, out.ctypes.data)
        return out

    def bootstrap_multi(self, cts, tables):
        """Every table of `tables` on every ciphertext of `cts` with ONE blind rotation per ciphertext.
        Returns [len(tables)][len(cts)][ct_words]."""
        cts = _c(cts, np.uint64).reshape(-1, self.ctw)
        tv0 = self.tv0()
        diffs = [self.build_tv_diff(t) for t in tables]
        out = np.empty((len(tables), len(cts), self.ctw), np.uint64)
        for i, ct in enumerate(cts):
            acc = self.blind_rotate(self.modswitch(self.keyswitch(ct)), tv0)
            for j, (d, post) in enumerate(diffs):
                out[j, i] = self.multi_extract(acc, d, post)
        return out

    def bootstrap_batch(self, cts, tables, table_ids=None, threads=0):
        """tables: list of int lists.  Returns (cts_out, threads_used)."""
        cts = _c(cts, np.uint64).reshape(-1, self.ctw)
        tvs = np.empty((len(tables), self.N), np.uint64)
        posts = np.empty(len(tables), np.uint64)
        for i, t in enumerate(tables):
            tvs[i], posts[i] = self.build_tv(t)
        ids = None if table_ids is None else _c(table_ids, np.uint32)
        out = np.empty_like(cts)
        used = lib().orc_bootstrap_batch(self._h, cts.ctypes.data, None if ids is None else ids.ctypes.data,
                                         tvs.ctypes.data, posts.ctypes.data, len(cts), out.ctypes.data, threads)
        return out, used


def polymul_schoolbook(a, b):
    a, b = _c(a, np.uint64), _c(b, np.uint64)
    c = np.empty_like(a)
    lib().orc_negacyclic_mul_schoolbook(a.ctypes.data, b.ctypes.data, c.ctypes.data, len(a))
    return c


def polymul_ntt(a, b):
    a, b = _c(a, np.uint64), _c(b, np.uint64)
    c = np.empty_like(a)
    lib().orc_negacyclic_mul_ntt(a.ctypes.data, b.ctypes.data, c.ctypes.data, int(np.log2(len(a))))
    return c
