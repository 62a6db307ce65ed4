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
    for bad in (Params(k=2), Params(log_n_poly=13), Params(l_bsk=5, beta_bsk=7), Params(p_msg=0)):
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
