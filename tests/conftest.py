import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the product library and the oracle are build artefacts (git-ignored); make sure they exist
    if not os.path.exists(os.path.join(ROOT, "tfhe_fbs_map_amd", "libfbsexec.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tfhe_fbs_map_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libtfhe_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def toy_params():
    from tfhe_fbs_map_amd import Params
    # small n so that the CPU oracle does a bootstrap in milliseconds; real N so the GPU kernels are the shipped ones
    return Params(n=12, log_n_poly=10, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=1 << 8)
