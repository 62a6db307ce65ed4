"""The whole-CU kernels reuse LDS words between the phases of a step (tools/soak_determinism.py says which); a race there would
show as a rare difference between two runs on the same inputs.  A short soak: every such launch shape, a few runs each, all
bit-identical and decrypting to the tables."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_whole_cu_kernels_are_deterministic():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_determinism.py"), "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "runs identical" in ln]
    assert len(lines) >= 28 and all("identical: True" in ln and "decrypts: True" in ln for ln in lines), r.stdout
    for kernel in ("k_blind_rotate_cu_pairs<11,1>", "k_blind_rotate_cu_pairs<11,2>", "k_blind_rotate_cu<10,3,2>", "k_blind_rotate_cu<10,3,2,lean>",
                   "k_blind_rotate_pairs_k2<10,4>", "k_blind_rotate_cu_k2", "k_blind_rotate_glwe<9,4,2,1>", "k_blind_rotate_glwe<9,4,2,2>",
                   "k_blind_rotate_glwe<9,4,2,3>"):
        assert any(kernel in ln for ln in lines), kernel
