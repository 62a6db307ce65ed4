"""The C-ABI shared library loads on a GPU-less host, exports exactly what include/fbs_exec.h declares,
and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fbs_exec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fbs_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from tfhe_fbs_map_amd import _native
    assert declared_symbols() == sorted(_native.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from tfhe_fbs_map_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_library_carries_gfx950_code_only():
    from tfhe_fbs_map_amd import _native
    blob = open(_native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"gfx1100"):
        assert other not in blob


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tfhe_fbs_map_amd import Context, FbsError, Params
    with pytest.raises(FbsError) as e:
        Context(Params())
    assert e.value.code in (-2, -1) and "HIP" in str(e.value) or "device" in str(e.value)


def test_bad_parameters_rejected_before_touching_the_device():
    from tfhe_fbs_map_amd import Context, FbsError, Params
    for bad in (Params(k=2, log_n_poly=11), Params(k=5, log_n_poly=9), Params(log_n_poly=13), Params(l_bsk=5, beta_bsk=7), Params(p_msg=0)):
        with pytest.raises(FbsError) as e:
            Context(bad)
        assert e.value.code == -1


def test_non_power_of_two_polynomial_size_is_refused_with_its_own_code():
    """BASELINE config 5 names a non-power-of-two N; DESIGN.md says why that ring is refused."""
    from tfhe_fbs_map_amd import FbsError, Params
    for n_poly in (1536, 768, 3 * 1024, 1000):
        with pytest.raises(FbsError) as e:
            Params.for_poly_size(n_poly)
        assert e.value.code == -5 and "not a power of two" in str(e.value)
    with pytest.raises(FbsError) as e:
        Params.for_poly_size(8192)
    assert e.value.code == -1
    assert Params.for_poly_size(2048, p_msg=31).log_n_poly == 11 and Params.for_poly_size(4096).log_n_poly == 12


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tfhe_fbs_map_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                # prose may mention the checker; code must not import, open or link it
                assert not re.search(r"(import\s+oracle|from\s+oracle|oracle[./]|libtfhe_oracle|tfhe_oracle|lut_oracle|orc_)", src), f


def _build_c_client(tmp_path):
    import subprocess
    exe = str(tmp_path / "capi_smoke")
    lib_dir = os.path.join(ROOT, "tfhe_fbs_map_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "capi_smoke.c"), "-o", exe,
                           "-L", lib_dir, "-lfbsexec", "-Wl,-rpath," + lib_dir])
    return exe


def test_header_is_plain_c_and_links(tmp_path):
    """A C11 program includes include/fbs_exec.h, links libfbsexec.so and calls it; without a GPU the library
    must answer FBS_E_DEVICE (exit code 3), not crash and not compute."""
    import subprocess
    import torch
    exe = _build_c_client(tmp_path)
    rc = subprocess.run([exe], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert rc.returncode == 0, rc.stdout + rc.stderr
    else:
        assert rc.returncode == 3, rc.stdout + rc.stderr


@pytest.mark.gpu
def test_c_client_bootstraps_on_the_gpu(tmp_path):
    import subprocess
    rc = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True)
    assert rc.returncode == 0 and rc.stdout.startswith("ok on gfx950"), rc.stdout + rc.stderr


def test_no_exception_crosses_the_boundary():
    """include/fbs_exec.h: "never throws, never aborts".  Every extern "C" entry point is a function-try-block; the test hook raises
    what a failing host allocation (std::vector / std::thread sized by caller input) or a library call would raise INSIDE one,
    and the call comes back with a code and a text -- in this very process, which goes on."""
    from tfhe_fbs_map_amd import _native
    lib = _native.lib
    want = {0: (-6, "out of host memory"), 1: (-6, "out of host memory"), 2: (-1, "internal error: raised on request"),
            3: (-1, "unknown internal error"), 9: (0, None)}
    for kind, (code, text) in want.items():
        assert lib.fbs_debug_raise(None, kind) == code, kind
        if text:
            assert text in lib.fbs_last_error(None).decode(), kind
    # and every entry point of the sources has the barrier: an `extern "C"` definition without `try` would be a hole
    for src in ("fbs_capi.cpp", "fbs_mapper_search.hip"):
        text = open(os.path.join(ROOT, "tfhe_fbs_map_amd", "csrc", src)).read()
        body = text[text.index('extern "C" {'):]
        for m in re.finditer(r"^(?:int|void|double|const char \*) ?(fbs_\w+)\(([^{};]*?)\) (try )?\{", body, flags=re.M):
            if m.group(1) in ("fbs_last_error", "fbs_device_info", "fbs_searcher_last_error", "fbs_searcher_last_kernel_ms"):
                continue                                    # one-line accessors of an existing string / number: nothing can throw
            assert m.group(3), "%s: %s has no function-try-block" % (src, m.group(1))


@pytest.mark.gpu
def test_absurd_counts_come_back_as_codes():
    """Counts are checked against FBS_MAX_* before anything is sized by them: a program description or a table set with
    corrupted counts returns FBS_E_INVALID (the arrays behind them are never touched), and the process survives."""
    import numpy as np
    from tfhe_fbs_map_amd import Params, _native as nat
    lib = nat.lib
    ctx = nat.Context(Params(n=8, log_n_poly=8, p_msg=7, sigma_lwe=1 << 6, sigma_glwe=1 << 4), seed=1)
    tv = ctx.tvset([[0, 1, 1, 0]])
    one8, one32, one64 = np.zeros(4, np.uint8), np.zeros(4, np.uint32), np.zeros(4, np.int64)

    def load(n_inputs, n_instr, n_terms, n_outputs, kind=one8):
        d = nat._ProgramDesc(n_inputs=n_inputs, n_instr=n_instr, n_terms=n_terms, n_outputs=n_outputs,
                             kind=None if kind is None else kind.ctypes.data, arg0=one32.ctypes.data, arg1=one32.ctypes.data,
                             const_coef=one64.ctypes.data, term_coef=one64.ctypes.data, term_src=one32.ctypes.data, out_wire=one64.ctypes.data)
        out = ctypes.c_void_p()
        return lib.fbs_program_load(ctx._h, ctypes.byref(d), tv._h, ctypes.byref(out)), out

    for args in ((0xFFFFFFFF, 2, 0, 0), (1, 0xFFFFFFF0, 0, 0), (2, 2, 0xF0000000, 0), (2, 2, 0, 0xFFFFFFFF), (1 << 27, (1 << 27) + 1, 0, 0)):
        rc, out = load(*args)
        assert rc == -1 and not out.value, args
        assert "program too large" in lib.fbs_last_error(ctx._h).decode()
    rc, _ = load(1, 2, 0, 0, kind=None)
    assert rc == -1 and "null array" in lib.fbs_last_error(ctx._h).decode()
    out = ctypes.c_void_p()
    assert lib.fbs_tvset_create(ctx._h, one32.ctypes.data, one32.ctypes.data, 0xFFFFFFFF, ctypes.byref(out)) == -1
    assert "FBS_MAX_TABLES" in lib.fbs_last_error(ctx._h).decode()
    # the context is as usable as before
    msgs = np.arange(4)
    assert np.array_equal(ctx.decrypt(ctx.bootstrap_batch(tv, ctx.encrypt(msgs, nonce0=1))), [0, 1, 1, 0])
    ctx.close()
