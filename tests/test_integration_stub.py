"""INTEGRATION.md section 2 is the text a reference maintainer would paste into fbs_mapper/fbs_exec_env.py (:208-229).  These tests
EXECUTE it: the fenced block is cut out of the document, bound as `eval` onto this package's `LutExecEnv` (it only reads
`self.instructions`, `self.outputs`, `self.stats()` and the node classes -- the reference's own surface), its parameter table is
held to the selector and to the noise model, and on the GPU it evaluates reference-mapped fixtures to the reference's goldens."""
import os
import re
import sys
import textwrap

import numpy as np
import pytest

from tests.helpers import assert_outputs_equal, load_fixture, subsample

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = doc[doc.index("## 2. The stub a reference maintainer would add"):doc.index("## 3. Entry points")]
    blocks = re.findall(r"```python\n(.*?)```", section, flags=re.S)
    assert len(blocks) == 1, "section 2 holds exactly one code block: the stub"
    return blocks[0]


def stub_table():
    src = stub_source()
    text = src[src.index("SETS = {"):src.index("# </stub-table>")]
    scope = {}
    exec(textwrap.dedent(text), scope)
    return scope["SETS"]


def bound_stub():
    """-> a subclass of the package's LutExecEnv whose `eval` is the stub's"""
    import ctypes
    from tfhe_fbs_map_amd import LutExecEnv as Real, _native
    src = stub_source()
    assert src.startswith("# fbs_mapper/fbs_exec_env.py")      # the comment line that says where the text goes (column 0)
    src = textwrap.dedent(src.split("\n", 1)[1])               # "    _fbs = None" / "    def eval" -> column 0
    scope = {}
    holder = type("LutExecEnv", (Real,), {})                   # the name the stub's body refers to
    scope["LutExecEnv"] = holder
    exec(src, scope)
    holder.eval = scope["eval"]
    holder._fbs = ctypes.CDLL(_native.LIB_PATH)                # the in-tree build (the stub itself asks the loader for "libfbsexec.so")
    return holder


def test_the_table_is_what_the_selector_returns_and_holds_six_sigma():
    from tfhe_fbs_map_amd import Params
    from tfhe_fbs_map_amd.params import margin_sigmas, security_bits
    from tfhe_fbs_map_amd.security import sigma_min
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_integration_stub_table as gen
    table = stub_table()
    assert table == gen.rows(), "INTEGRATION.md is stale: python tools/make_integration_stub_table.py --write"
    assert gen.table_text() in open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = stub_source()
    sig = eval(re.search(r"sig = (lambda dim: .*)\n", src).group(1), {"math": __import__("math"), "q": 0x3FFFFFF84001})
    for (p, norm2), (n, log_n, k, l, beta, t, gamma, group) in table.items():
        assert sig(n) == sigma_min(n) and sig(k << log_n) == sigma_min(k << log_n)      # the stub's noise IS the security floor
        prm = Params(n=n, log_n_poly=log_n, k=k, l_bsk=l, beta_bsk=beta, t_ksk=t, gamma_ksk=gamma, p_msg=p, sigma_lwe=sig(n),
                     sigma_glwe=sig(k << log_n), bsk_group=group)
        assert margin_sigmas(prm, norm2) >= 6.0, (p, norm2, margin_sigmas(prm, norm2))
        assert security_bits(prm) >= 127.9, (p, norm2)
        # ... and so for every program the row admits (a smaller p widens the box, a smaller norm shrinks the noise)
        for pp, nn in ((2, 1), (p, 1), (max(2, p // 2), norm2)):
            assert margin_sigmas(prm.replace(p_msg=pp), nn) >= 6.0
    bounds = sorted(table)
    assert bounds[0][0] >= 4 and bounds[-1][0] >= 31                 # the reference's p = 4 point and BASELINE's fbs_size = 31


def test_the_stub_compiles_and_reads_only_the_reference_surface():
    cls = bound_stub()
    assert callable(cls.eval)
    src = stub_source()
    used = set(re.findall(r"self\.(\w+)", src))
    assert used <= {"instructions", "outputs", "stats"}, used        # fbs_exec_env.py:65-69, :245
    assert set(re.findall(r"LutExecEnv\.(\w+)", src)) <= {"Input", "LinearProd", "Const", "_fbs"}    # :22-49


@pytest.mark.gpu
@pytest.mark.parametrize("name,fbs_size,T", [("demo_fbs_exec_env", 4, 2), ("full_adder__search_p7", 7, 24), ("adder8__basic_p2", 4, 16),
                                             ("aes_sbox__search_p3", 3, 16), ("mul4__search_p15", 15, 8)])
def test_the_stub_evaluates_reference_programs_on_the_gpu(name, fbs_size, T):
    """The pasted method, run: same dict in, same dict out as the reference's cleartext `eval` (its goldens)."""
    from tfhe_fbs_map_amd import parse_fbs
    rec = load_fixture(name)
    ins, expect = subsample(rec, T)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    cls = bound_stub()
    env.__class__ = cls                                              # same object, the stub's eval
    got = env.eval(ins, fbs_size=fbs_size, seed=7)
    assert_outputs_equal(got, expect)
    got2 = env.eval(ins, fbs_size=fbs_size)                          # the default: 32 bytes from the OS
    assert_outputs_equal(got2, expect)
