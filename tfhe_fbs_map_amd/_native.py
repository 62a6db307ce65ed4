"""ctypes binding of libfbsexec.so (C ABI: include/fbs_exec.h).

There is deliberately no fallback: if the shared library is missing or cannot be
loaded, importing this module raises, and if no gfx950 GPU is present
`Context(...)` raises `FbsError` -- the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, asdict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FBS_LIB") or os.path.join(_HERE, "libfbsexec.so")   # FBS_LIB: kernel-variant experiments

from .security import MODULUS, MODULUS_BITS, sigma_min      # noqa: E402,F401


RANDOMNESS_GRADE = ("test-grade: ChaCha20 streams keyed by the context seed; noise = integer Irwin-Hall(12) stand-in for a discrete "
                    "Gaussian, bounded at 6 sigma.  Bring keys made with a production sampler through Context.import_keys")


class FbsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libfbsexec error {code}: {text}")
        self.code = code


class _Params(C.Structure):
    _fields_ = [(f, C.c_uint32) for f in
                ("n", "log_n_poly", "k", "l_bsk", "beta_bsk", "t_ksk", "gamma_ksk", "p_msg")] + \
               [("sigma_lwe", C.c_uint64), ("sigma_glwe", C.c_uint64), ("bsk_group", C.c_uint32), ("reserved", C.c_uint32)]


class _Layout(C.Structure):
    _fields_ = [(f, C.c_uint32) for f in ("n_slots", "n_levels", "max_width", "max_sources", "n_bootstrap", "n_keyswitch",
                                          "n_inputs", "n_outputs", "n_rotations", "row_words")]


class _ProgramDesc(C.Structure):
    _fields_ = [("n_inputs", C.c_uint32), ("n_instr", C.c_uint32), ("n_terms", C.c_uint32),
                ("n_outputs", C.c_uint32),
                ("kind", C.c_void_p), ("arg0", C.c_void_p), ("arg1", C.c_void_p),
                ("const_coef", C.c_void_p), ("term_coef", C.c_void_p), ("term_src", C.c_void_p),
                ("out_wire", C.c_void_p)]


@dataclass(frozen=True)
class Params:
    """Cryptographic parameter set.  The shape defaults to BASELINE.md's synthetic set (n=630 N=1024 k=1 l=3 beta=7
    t=8 gamma=2).  A noise left at None becomes the smallest standard deviation that is 128-bit secure at its
    dimension (`security.sigma_min`); anything lower is an explicit choice -- `reduced_noise()` is the benchmark
    setting (2^-40 q, NOT secure), `params.P1024` the benchmark set built with it, `params.choose_params` the
    selector that returns secure AND correct sets."""
    n: int = 630
    log_n_poly: int = 10
    k: int = 1
    l_bsk: int = 3
    beta_bsk: int = 7
    t_ksk: int = 8
    gamma_ksk: int = 2
    p_msg: int = 15
    sigma_lwe: int | None = None      # key-switching-key noise, absolute units of 1/q
    sigma_glwe: int | None = None     # bootstrapping-key and fresh-input noise
    bsk_group: int = 1                # key bits per blind-rotation step: 1, or 2 (n/2 steps on bundles of 3 GGSW samples)

    def __post_init__(self):
        if self.sigma_lwe is None:
            object.__setattr__(self, "sigma_lwe", sigma_min(self.n))
        if self.sigma_glwe is None:
            object.__setattr__(self, "sigma_glwe", sigma_min(self.k * (1 << self.log_n_poly)))

    @classmethod
    def for_poly_size(cls, poly_size: int, **kw):
        """Parameter set for polynomial size N.  A non-power-of-two N (BASELINE config 5 names one) raises
        FbsError(FBS_E_POLY_SIZE): see include/fbs_exec.h `fbs_poly_size_check` for why that ring is refused."""
        rc = lib.fbs_poly_size_check(int(poly_size))
        if rc != 0:
            raise FbsError(rc, lib.fbs_last_error(None).decode())
        return cls(log_n_poly=int(poly_size).bit_length() - 1, **kw)

    def reduced_noise(self, sigma: int = 1 << 6):
        """The same shape with both noises at `sigma` (default 2^6 = 2^-40 q): throughput benchmarks and parity tests
        only -- far below what any security level needs at these dimensions."""
        return self.replace(sigma_lwe=sigma, sigma_glwe=sigma)

    @property
    def N(self):
        return 1 << self.log_n_poly

    @property
    def big_dim(self):
        return self.k * self.N

    @property
    def ct_words(self):
        return self.big_dim + 1

    def replace(self, **kw):
        d = asdict(self)
        d.update(kw)
        return Params(**d)

    def to_c(self):
        return _Params(reserved=0, **asdict(self))

    def bytes_per_fbs(self):
        """Algorithmic bytes one FBS must consume: every
        bootstrapping-key row and key-switching-key row once, its input and
        output ciphertext and its test vector."""
        N, k, n = self.N, self.k, self.n
        ggsw = (k + 1) * self.l_bsk * (k + 1) * N * 8
        bsk = (n // 2 * 3 if self.bsk_group == 2 else n) * ggsw
        ksk = k * N * self.t_ksk * (n + 1) * 8
        return bsk + ksk + 2 * (k * N + 1) * 8 + N * 8


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  tfhe_fbs_map_amd has no CPU fallback.")
    # PyTorch ships its own copy of the HIP runtime; if it is going to be used in this process (device
    # tensors, RCCL) it has to be the first one loaded, or torch later finds "No HIP GPUs".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, sz, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int
    sig = {
        "fbs_poly_size_check": (i32, [u32]),
        "fbs_ctx_create": (i32, [C.POINTER(_Params), u64, i32, C.POINTER(vp)]),
        "fbs_ctx_create_seeded": (i32, [C.POINTER(_Params), vp, i32, C.POINTER(vp)]),
        "fbs_ctx_destroy": (None, [vp]),
        "fbs_ctx_reserve": (i32, [vp, sz, sz, sz]),
        "fbs_ctx_tune": (i32, [vp, C.c_char_p, C.c_int64]),
        "fbs_ctx_stat": (i32, [vp, C.c_char_p, C.POINTER(C.c_int64)]),
        "fbs_last_error": (C.c_char_p, [vp]),
        "fbs_device_info": (C.c_char_p, [vp]),
        "fbs_keygen": (i32, [vp]),
        "fbs_key_sizes": (i32, [vp, C.POINTER(sz * 4)]),
        "fbs_export_keys": (i32, [vp, vp, vp, vp, vp]),
        "fbs_import_keys": (i32, [vp, vp, vp, vp, vp]),
        "fbs_encrypt": (i32, [vp, vp, sz, u64, vp]),
        "fbs_encrypt_fresh": (i32, [vp, vp, sz, vp, C.POINTER(u64)]),
        "fbs_decrypt": (i32, [vp, vp, sz, vp]),
        "fbs_tvset_create": (i32, [vp, vp, vp, u32, C.POINTER(vp)]),
        "fbs_tvset_destroy": (None, [vp]),
        "fbs_bootstrap_batch": (i32, [vp, vp, vp, vp, sz, vp]),
        "fbs_bootstrap_batch_dev": (i32, [vp, vp, vp, vp, sz, vp, vp]),
        "fbs_lincomb_dev": (i32, [vp, vp, sz, u32, vp, vp, vp, vp, vp, vp]),
        "fbs_bootstrap_wires_dev": (i32, [vp, vp, vp, sz, u32, vp, vp, vp, sz, sz, vp]),
        "fbs_program_load": (i32, [vp, C.POINTER(_ProgramDesc), vp, C.POINTER(vp)]),
        "fbs_program_load_ex": (i32, [vp, C.POINTER(_ProgramDesc), vp, u32, C.POINTER(vp)]),
        "fbs_table_fusion_norms": (i32, [vp, u32, C.POINTER(u64), C.POINTER(u64)]),
        "fbs_program_destroy": (None, [vp]),
        "fbs_program_info": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "fbs_eval": (i32, [vp, vp, vp, sz, vp]),
        "fbs_eval_dev": (i32, [vp, vp, vp, sz, vp, vp]),
        "fbs_program_layout": (i32, [vp, C.POINTER(_Layout)]),
        "fbs_program_level": (i32, [vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        "fbs_program_io_slots": (i32, [vp, vp, vp]),
        "fbs_level_lincomb_dev": (i32, [vp, vp, u32, vp, sz, sz, sz, vp]),
        "fbs_level_bootstrap_dev": (i32, [vp, vp, u32, vp, sz, sz, sz, sz, sz, vp, vp]),
        "fbs_level_scatter_dev": (i32, [vp, vp, u32, vp, sz, sz, sz, vp, sz, sz, vp]),
        "fbs_profile_enable": (i32, [vp, i32]),
        "fbs_profile_read": (i32, [vp, C.POINTER(C.c_double * 3), C.POINTER(u64 * 3), i32]),
        "fbs_profile_kernel": (C.c_char_p, [vp, i32]),
        "fbs_kernel_catalog": (C.c_char_p, []),
        "fbs_profile_kernels": (i32, [vp, vp, sz, C.POINTER(sz)]),
        "fbs_sync": (i32, [vp, vp]),
        "fbs_debug_polymul": (i32, [vp, vp, vp, vp]),
        "fbs_debug_raise": (i32, [vp, i32]),
        "fbs_searcher_create": (i32, [i32, C.POINTER(vp)]),
        "fbs_searcher_destroy": (None, [vp]),
        "fbs_searcher_last_error": (C.c_char_p, [vp]),
        "fbs_searcher_last_kernel_ms": (C.c_double, [vp]),
        "fbs_search_lincomb_coefs": (i32, [vp, vp, vp, vp, u32, u32, u32, vp, vp, C.POINTER(i32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = the library does not match the header
        fn.restype = res
        fn.argtypes = args
    return lib


EXPORTED_SYMBOLS = (
    "fbs_poly_size_check", "fbs_ctx_create", "fbs_ctx_create_seeded", "fbs_ctx_reserve", "fbs_ctx_tune", "fbs_ctx_stat",
    "fbs_import_keys", "fbs_encrypt_fresh", "fbs_ctx_destroy", "fbs_last_error", "fbs_device_info", "fbs_keygen",
    "fbs_key_sizes", "fbs_export_keys", "fbs_encrypt", "fbs_decrypt", "fbs_tvset_create",
    "fbs_tvset_destroy", "fbs_bootstrap_batch", "fbs_bootstrap_batch_dev", "fbs_lincomb_dev",
    "fbs_bootstrap_wires_dev", "fbs_program_load", "fbs_program_load_ex", "fbs_table_fusion_norms", "fbs_program_destroy",
    "fbs_program_info",
    "fbs_searcher_create", "fbs_searcher_destroy", "fbs_searcher_last_error", "fbs_searcher_last_kernel_ms",
    "fbs_search_lincomb_coefs", "fbs_eval", "fbs_eval_dev", "fbs_program_layout", "fbs_program_level", "fbs_program_io_slots",
    "fbs_level_lincomb_dev", "fbs_level_bootstrap_dev", "fbs_level_scatter_dev", "fbs_profile_enable", "fbs_profile_kernel", "fbs_kernel_catalog", "fbs_profile_kernels", "fbs_profile_read", "fbs_sync", "fbs_debug_polymul", "fbs_debug_raise",
)

lib = _load()


def _ptr(a):
    return None if a is None else a.ctypes.data


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def kernel_catalog():
    """Names of every kernel instantiation the launchers can pick (fbs_kernel_catalog)."""
    return [n for n in lib.fbs_kernel_catalog().decode().split("\n") if n]


class TvSet:
    def __init__(self, ctx, tables):
        self.ctx = ctx
        self.tables = [list(map(int, t)) for t in tables]
        vals = _c([v for t in self.tables for v in t] or [0], np.int32)
        off = np.zeros(len(self.tables) + 1, np.uint32)
        off[1:] = np.cumsum([len(t) for t in self.tables])
        h = C.c_void_p()
        ctx._check(lib.fbs_tvset_create(ctx._h, _ptr(vals), _ptr(off), len(self.tables), C.byref(h)))
        self._h = h

    def fusion_norms(self, table):
        """(|D_F|^2, |G_F|^2) of table `table` (include/fbs_exec.h, fbs_table_fusion_norms): what sharing a blind rotation
        does to its output noise."""
        d, g = C.c_uint64(), C.c_uint64()
        self.ctx._check(lib.fbs_table_fusion_norms(self._h, table, C.byref(d), C.byref(g)))
        return d.value, g.value

    def __del__(self):
        if getattr(self, "_h", None) and self.ctx._h and lib is not None:
            lib.fbs_tvset_destroy(self._h)
            self._h = None


class Program:
    FUSE_TABLES = 1    # FBS_LOAD_FUSE_TABLES

    def __init__(self, ctx, tvset, n_inputs, kind, arg0, arg1, const_coef, term_coef, term_src, out_wire, fuse_tables=False):
        """fuse_tables: the tables of a source that several Bootstraps read share ONE blind rotation
        (include/fbs_exec.h, FBS_LOAD_FUSE_TABLES)."""
        self.ctx, self.tvset = ctx, tvset
        self.fused = bool(fuse_tables)
        self._keep = [_c(kind, np.uint8), _c(arg0, np.uint32), _c(arg1, np.uint32), _c(const_coef, np.int64),
                      _c(term_coef, np.int64), _c(term_src, np.uint32), _c(out_wire, np.int64)]
        k = self._keep
        desc = _ProgramDesc(n_inputs, len(k[0]), len(k[4]), len(k[6]), *[_ptr(a) for a in k])
        h = C.c_void_p()
        ctx._check(lib.fbs_program_load_ex(ctx._h, C.byref(desc), tvset._h, self.FUSE_TABLES if fuse_tables else 0, C.byref(h)))
        self._h = h
        self.n_inputs, self.n_outputs = n_inputs, len(k[6])
        lay = _Layout()
        ctx._check(lib.fbs_program_layout(h, C.byref(lay)))
        self.depth, self.max_width, self.n_bootstrap = lay.n_levels, lay.max_width, lay.n_bootstrap
        self.n_slots, self.n_keyswitch, self.max_sources = lay.n_slots, lay.n_keyswitch, lay.max_sources
        self.n_rotations = lay.n_rotations
        self.row_words = lay.row_words          # words per row of the d_rows arrays of the level calls (2N for a fused program)
        self.in_slot = np.empty(self.n_inputs, np.uint32)
        self.out_slot = np.empty(self.n_outputs, np.int64)
        ctx._check(lib.fbs_program_io_slots(h, _ptr(self.in_slot), _ptr(self.out_slot)))
        self.level_width, self.level_sources = [], []
        for L in range(self.depth):
            a, b = C.c_uint32(), C.c_uint32()
            ctx._check(lib.fbs_program_level(h, L, C.byref(a), C.byref(b)))
            self.level_width.append(a.value)
            self.level_sources.append(b.value)

    def eval(self, in_cts, T):
        ctw = self.ctx.params.ct_words
        in_cts = _c(in_cts, np.uint64).reshape(self.n_inputs, T, ctw)
        out = np.empty((self.n_outputs, T, ctw), np.uint64)
        self.ctx._check(lib.fbs_eval(self.ctx._h, self._h, _ptr(in_cts), T, _ptr(out)))
        return out

    # device-pointer entry points (ints from torch.Tensor.data_ptr()); asynchronous on `stream`, no host copies
    def eval_dev(self, d_in, T, d_out, stream=0):
        self.ctx._check(lib.fbs_eval_dev(self.ctx._h, self._h, d_in or None, T, d_out or None, stream or None))

    def level_lincomb_dev(self, level, d_wires, T, s_begin, s_count, stream=0):
        self.ctx._check(lib.fbs_level_lincomb_dev(self.ctx._h, self._h, level, d_wires, T, s_begin, s_count, stream or None))

    def level_bootstrap_dev(self, level, d_wires, T, s_begin, s_count, f_begin, f_end, d_rows=0, stream=0):
        self.ctx._check(lib.fbs_level_bootstrap_dev(self.ctx._h, self._h, level, d_wires, T, s_begin, s_count, f_begin, f_end,
                                                    d_rows or None, stream or None))

    def level_scatter_dev(self, level, d_wires, T, s_begin, s_count, d_rows, f_begin, f_end, stream=0):
        self.ctx._check(lib.fbs_level_scatter_dev(self.ctx._h, self._h, level, d_wires, T, s_begin, s_count, d_rows, f_begin,
                                                  f_end, stream or None))

    def close(self):
        if getattr(self, "_h", None) and self.ctx._h and lib is not None:
            lib.fbs_program_destroy(self._h)
        self._h = None

    def __del__(self):
        self.close()


class Context:
    """One GPU, one parameter set, one key set."""

    def __init__(self, params: Params, seed: int | bytes | None = None, device: int = 0, keygen: bool = True):
        """seed: what all key material and encryption randomness derive from.
        * an int: the REPRODUCIBLE form (fbs_ctx_create, 64 bits) -- tests and benchmarks pass a constant so that the CPU
          oracle can be keyed identically.  Not a way to make production keys.
        * None or 32 bytes: fbs_ctx_create_seeded -- 256 bits (None: from os.urandom) with the parameter set mixed into the
          derivation.
        Either way the noise sampler is a test-grade stand-in for a discrete Gaussian (`RANDOMNESS_GRADE`); a deployment
        that needs more brings its own keys with `import_keys`.  keygen=False leaves the context without keys (for
        `import_keys`)."""
        self.params = params
        self.seed = seed
        self._h = C.c_void_p()
        cp = params.to_c()
        if isinstance(seed, int):
            rc = lib.fbs_ctx_create(C.byref(cp), seed, device, C.byref(self._h))
        else:
            raw = os.urandom(32) if seed is None else bytes(seed)
            if len(raw) != 32:
                raise ValueError("a byte seed has 32 bytes")
            rc = lib.fbs_ctx_create_seeded(C.byref(cp), raw, device, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise FbsError(rc, lib.fbs_last_error(None).decode())
        if keygen:
            self.keygen()

    def _check(self, rc):
        if rc != 0:
            raise FbsError(rc, lib.fbs_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and lib is not None:      # `lib` is None while the interpreter shuts down
            lib.fbs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def device_info(self):
        return lib.fbs_device_info(self._h).decode()

    def keygen(self):
        self._check(lib.fbs_keygen(self._h))

    def export_keys(self):
        sizes = (C.c_size_t * 4)()
        self._check(lib.fbs_key_sizes(self._h, C.byref(sizes)))
        arrs = [np.empty(sizes[i], np.uint64) for i in range(4)]
        self._check(lib.fbs_export_keys(self._h, *[_ptr(a) for a in arrs]))
        return dict(sk_lwe=arrs[0], sk_glwe=arrs[1], bsk=arrs[2], ksk=arrs[3])

    def import_keys(self, sk_lwe, sk_glwe, bsk, ksk):
        """Keys made elsewhere (layout of `export_keys`) instead of `keygen`: a caller's own CSPRNG and sampler, or a checker's."""
        arrs = [_c(a, np.uint64).ravel() for a in (sk_lwe, sk_glwe, bsk, ksk)]
        sizes = (C.c_size_t * 4)()
        self._check(lib.fbs_key_sizes(self._h, C.byref(sizes)))
        for a, want, name in zip(arrs, sizes, ("sk_lwe", "sk_glwe", "bsk", "ksk")):
            if a.size != want:
                raise ValueError(f"{name} has {a.size} words, the parameter set needs {want}")
        self._check(lib.fbs_import_keys(self._h, *[_ptr(a) for a in arrs]))

    def reserve(self, max_keyswitches=0, max_shared_rows=0, wire_words=0):
        """Size the scratch up front so that no later `*_dev` call has to grow it (growing blocks): include/fbs_exec.h."""
        self._check(lib.fbs_ctx_reserve(self._h, int(max_keyswitches), int(max_shared_rows), int(wire_words)))

    def tune(self, **knobs):
        """Launcher knobs (`fbs_ctx_tune`): which kernel shape a launch takes; results never depend on them."""
        for k, v in knobs.items():
            self._check(lib.fbs_ctx_tune(self._h, k.encode(), int(v)))

    def stat(self, name):
        v = C.c_int64()
        self._check(lib.fbs_ctx_stat(self._h, name.encode(), C.byref(v)))
        return v.value

    def encrypt(self, msgs, nonce0=None):
        """nonce0=None: streams nobody has used (the context counts them: no two calls share mask or noise); an int: ciphertext
        i takes stream nonce0 + i -- reproducible, for tests and checkers."""
        msgs = _c(msgs, np.int64)
        cts = np.empty(msgs.shape + (self.params.ct_words,), np.uint64)
        if nonce0 is None:
            self._check(lib.fbs_encrypt_fresh(self._h, _ptr(msgs), msgs.size, _ptr(cts), None))
        else:
            self._check(lib.fbs_encrypt(self._h, _ptr(msgs), msgs.size, nonce0, _ptr(cts)))
        return cts

    def decrypt(self, cts):
        cts = _c(cts, np.uint64)
        out = np.empty(cts.shape[:-1], np.int64)
        self._check(lib.fbs_decrypt(self._h, _ptr(cts), out.size, _ptr(out)))
        return out

    def tvset(self, tables):
        return TvSet(self, tables)

    def bootstrap_batch(self, tvset, cts, table_ids=None):
        cts = _c(cts, np.uint64)
        count = cts.size // self.params.ct_words
        ids = None if table_ids is None else _c(table_ids, np.uint32)
        out = np.empty_like(cts)
        self._check(lib.fbs_bootstrap_batch(self._h, tvset._h, _ptr(cts), _ptr(ids), count, _ptr(out)))
        return out

    # device-pointer entry points (ints from torch.Tensor.data_ptr()); asynchronous on `stream`
    def bootstrap_batch_dev(self, tvset, d_in, d_table_ids, count, d_out, stream=0):
        self._check(lib.fbs_bootstrap_batch_dev(self._h, tvset._h, d_in, d_table_ids or None, count, d_out,
                                                stream or None))

    def lincomb_dev(self, d_wires, T, dst, term_off, srcs, coefs, consts, stream=0):
        dst, term_off, srcs = _c(dst, np.uint32), _c(term_off, np.uint32), _c(srcs, np.uint32)
        coefs, consts = _c(coefs, np.int64), _c(consts, np.int64)
        self._check(lib.fbs_lincomb_dev(self._h, d_wires, T, len(dst), _ptr(dst), _ptr(term_off), _ptr(srcs),
                                        _ptr(coefs), _ptr(consts), stream or None))

    def bootstrap_wires_dev(self, tvset, d_wires, T, src, dst, table_ids, s_begin, s_end, stream=0):
        src, dst, table_ids = _c(src, np.uint32), _c(dst, np.uint32), _c(table_ids, np.uint32)
        self._check(lib.fbs_bootstrap_wires_dev(self._h, tvset._h, d_wires, T, len(src), _ptr(src), _ptr(dst),
                                                _ptr(table_ids), s_begin, s_end, stream or None))

    def profile(self, on=True):
        self._check(lib.fbs_profile_enable(self._h, int(on)))

    def profile_read(self, reset=True):
        ms = (C.c_double * 3)()
        cnt = (C.c_uint64 * 3)()
        self._check(lib.fbs_profile_read(self._h, C.byref(ms), C.byref(cnt), int(reset)))
        names = ("keyswitch", "blind_rotate", "lincomb")
        return {n: dict(ms=ms[i], launches=int(cnt[i]), kernel=lib.fbs_profile_kernel(self._h, i).decode())
                for i, n in enumerate(names)}

    def profile_kernels(self):
        """{kernel instantiation: dict(kind, launches, ms)} since the last reset -- a launch cut into a whole-round part and a
        remainder shows as two entries."""
        need = C.c_size_t()
        self._check(lib.fbs_profile_kernels(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        self._check(lib.fbs_profile_kernels(self._h, buf, need.value, None))
        out = {}
        for line in buf.value.decode().splitlines():
            kind, name, launches, ms = line.split("\t")
            out[name] = dict(kind=("keyswitch", "blind_rotate", "lincomb")[int(kind)], launches=int(launches), ms=float(ms))
        return out

    def sync(self, stream=0):
        self._check(lib.fbs_sync(self._h, stream or None))

    def debug_polymul(self, a, b):
        a, b = _c(a, np.uint64), _c(b, np.uint64)
        c = np.empty_like(a)
        self._check(lib.fbs_debug_polymul(self._h, _ptr(a), _ptr(b), _ptr(c)))
        return c
