"""Gate-level front door (tfhe_fbs_map_amd/netlist.py) against what the reference wrote for the same circuits:
tests/golden/_netlists.json.gz holds, per circuit, the reference's BLIF text, printed netlist, stats, seed-42
inputs with `BitExecEnv.eval` outputs and the program its `MapToFBSBasic` prints (capture_reference.py)."""
import gzip
import io
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, _decode
from tfhe_fbs_map_amd.netlist import BitExecEnv, map_basic, parse_blif, parse_bristol

with gzip.open(os.path.join(GOLDEN, "_netlists.json.gz"), "rb") as _f:
    NETLISTS = json.loads(_f.read().decode())
CASES = sorted(NETLISTS)


def _printed(obj, **kw):
    buf = io.StringIO()
    obj.print(os=buf, **kw)
    return buf.getvalue()


def _inputs(rec):
    return {k: _decode(v) for k, v in rec["inputs"].items()}


def _same_outputs(got, rec):
    exp = {k: _decode(v) for k, v in rec["outputs"].items()}
    assert set(map(str, got.keys())) == set(exp.keys())
    for k, v in got.items():
        e = exp[str(k)]
        if isinstance(e, int):
            assert int(np.asarray(v).reshape(-1)[0]) == e
        else:
            assert np.array_equal(np.broadcast_to(np.asarray(v), e.shape), e), k


@pytest.mark.parametrize("case", CASES)
def test_blif_reader_reproduces_the_reference_netlist(case):
    rec = NETLISTS[case]
    if _output_shadows_input(rec):
        # e.g. ascon_lut: outputs x0..x4 carry the inputs' names, so the BLIF the reference writes drives x0 twice
        with pytest.raises(AssertionError, match="driven twice"):
            parse_blif(rec["blif"])
        return
    env = parse_blif(rec["blif"])
    _same_outputs(env.eval(_inputs(rec)), rec)
    # the BLIF the reference writes names output buffers; up to those the gate list is the reference's
    assert [i.name for i in env.inputs] == rec["blif"].splitlines()[1].split()[1:]
    ref_gates = [ln for ln in rec["print"].splitlines() if " = " in ln and not ln.startswith("Output")]
    n_buffers = rec["blif"].count("\n1 1\n")
    assert len(env.instructions) == len(ref_gates) + n_buffers


def _output_shadows_input(rec):
    lines = rec["blif"].splitlines()
    return bool(set(lines[1].split()[1:]) & set(lines[2].split()[1:]))


@pytest.mark.parametrize("case", [c for c in CASES if not _output_shadows_input(NETLISTS[c])])
def test_blif_round_trip_is_a_fixed_point(case):
    env = parse_blif(NETLISTS[case]["blif"])
    first = io.StringIO()
    env.to_blif(fs=first, model_name=case)
    again = io.StringIO()
    parse_blif(first.getvalue()).to_blif(fs=again, model_name=case)
    assert first.getvalue() == again.getvalue()


@pytest.mark.parametrize("case", CASES)
def test_basic_lowering_matches_the_reference_program(case):
    """Same gates in the same order give, line for line, the program MapToFBSBasic printed."""
    rec = NETLISTS[case]
    env = _rebuild_like_reference(rec)
    assert _printed(env) == rec["print"]
    assert env.stats() == rec["stats"]
    lut = map_basic(env)
    assert _printed(lut, show_outputs=True) == rec["basic_fbs"]
    from oracle import lut_oracle
    _same_outputs(lut_oracle.eval_fbs_text(rec["basic_fbs"], _inputs(rec)), rec)


def _rebuild_like_reference(rec):
    """Re-create the circuit through the builder API from the reference's printed netlist (names included)."""
    import ast
    import re
    env = BitExecEnv()
    nodes = {"0": BitExecEnv.CONST0, "1": BitExecEnv.CONST1}
    for ln in rec["print"].splitlines():
        if ln.startswith("Output "):
            name, src = ln[len("Output "):].split(" = ")
            env.output(name, nodes[src])
            continue
        name, rhs = ln.split(" = ", 1)
        m = re.fullmatch(r"(\w+)\((.*)\)", rhs)
        kind, args = m.group(1), m.group(2)
        if kind == "Input":
            nodes[name] = env.input(name)
        elif kind == "LUT":
            ins, table = re.fullmatch(r"\[(.*?)\], (\[.*\])", args).groups()
            nodes[name] = env.op_lut([nodes[s] for s in ins.split(", ")], ast.literal_eval(table), name=name)
        else:
            srcs = [nodes[s] for s in args.split(", ")]
            op = {"AND": env.op_and, "XOR": env.op_xor, "OR": env.op_or, "Not": env.op_not}[kind]
            nodes[name] = op(*srcs, name=name)
    return env


def test_cover_semantics():
    text = """
    # a comment
    .model t
    .inputs a b \\
            c
    .outputs x y z k1 k0 w
    .names a b c x      # off-set cover, don't-cares
    1-0 0
    01- 0
    .names a b y
    11 1
    .names y z
    0 1
    .names k1
    1
    .names k0
    .names x y z w
    1-- 1
    -11 1
    .end
    """
    env = parse_blif(text)
    T = 8
    ins = {"a": np.array([0, 0, 0, 0, 1, 1, 1, 1]), "b": np.array([0, 0, 1, 1, 0, 0, 1, 1]), "c": np.array([0, 1] * 4)}
    out = env.eval(ins)
    a, b, c = ins["a"], ins["b"], ins["c"]
    x = 1 - ((a & (1 - c)) | ((1 - a) & b))
    y = a & b
    z = 1 - y
    assert np.array_equal(out["x"], x) and np.array_equal(out["y"], y) and np.array_equal(out["z"], z)
    assert int(np.asarray(out["k1"]).reshape(-1)[0]) == 1 and int(np.asarray(out["k0"]).reshape(-1)[0]) == 0
    assert np.array_equal(out["w"], x | (y & z))
    assert env.outputs["k1"] is BitExecEnv.CONST1 and env.outputs["k0"] is BitExecEnv.CONST0
    assert T == len(out["x"])


def test_blif_use_before_definition_and_errors():
    env = parse_blif(".model m\n.inputs a b\n.outputs o\n.names t a o\n11 1\n.names a b t\n01 1\n10 1\n.end\n")
    out = env.eval({"a": [0, 0, 1, 1], "b": [0, 1, 0, 1]})
    assert list(out["o"]) == [0, 0, 1, 0]
    with pytest.raises(AssertionError):
        parse_blif(".model m\n.inputs a\n.outputs o\n.names a q o\n11 1\n.end\n")          # q undefined
    with pytest.raises(AssertionError):
        parse_blif(".model m\n.inputs a\n.outputs o\n.names a o\n1 1\n0 0\n.end\n")          # mixed phases
    with pytest.raises(AssertionError):
        parse_blif(".model m\n.inputs a\n.outputs o\n.names o p\n1 1\n.names p o\n1 1\n.end\n")   # loop
    with pytest.raises(AssertionError):
        parse_blif(".model m\n.inputs a\n.outputs o\n.latch a o 0\n.end\n")


def test_constant_folding_of_the_builder():
    env = BitExecEnv()
    a = env.input("a")
    assert env.op_and(a, env.CONST0) is env.CONST0 and env.op_and(env.CONST1, a) is a
    assert env.op_or(a, env.CONST1) is env.CONST1 and env.op_or(env.CONST0, a) is a
    assert env.op_xor(a, env.CONST0) is a
    n = env.op_xor(env.CONST1, a)
    assert str(n) == "Not(a)" and env.op_not(env.CONST0) is env.CONST1
    with pytest.raises(AssertionError):
        env.op_lut([a], [0, 0])
    with pytest.raises(AssertionError):
        env.op_lut([a, a], [0, 1])
    with pytest.raises(AssertionError):
        env.op_and(a, a)


BRISTOL_ADDER = """8 14
2 2 2
1 3

2 1 0 2 4 XOR
2 1 0 2 5 AND
2 1 1 3 6 XOR
2 1 6 5 7 XOR
2 1 1 3 8 AND
2 1 6 5 9 AND
2 1 8 9 10 XOR
1 1 4 11 EQW
1 1 7 12 EQW
1 1 10 13 EQW
"""


def test_bristol_two_bit_adder():
    text = BRISTOL_ADDER.replace("8 14", "10 14")
    env = parse_bristol(text)
    assert [i.name for i in env.inputs] == ["i_0", "i_1", "i_2", "i_3"]
    vals = {"i_%d" % k: np.array([(s >> k) & 1 for s in range(16)]) for k in range(4)}
    out = env.eval(vals)
    for s in range(16):
        x = (s & 1) | ((s >> 1) & 1) << 1
        y = ((s >> 2) & 1) | ((s >> 3) & 1) << 1
        total = x + y
        assert [int(out[11][s]), int(out[12][s]), int(out[13][s])] == [total & 1, (total >> 1) & 1, total >> 2]
    lut = map_basic(env)
    got = lut.stats()
    assert got["nb_bootstrap"] == 7


def test_bristol_inv_and_old_header():
    env = parse_bristol("2 4\n1 1 1\n2 1 0 1 2 AND\n1 1 2 3 INV\n")
    out = env.eval({"i_0": [0, 0, 1, 1], "i_1": [0, 1, 0, 1]})
    assert list(out[3]) == [1, 1, 1, 0]


def test_mapped_netlist_runs_in_the_cleartext_oracle():
    """BLIF -> map_basic -> .fbs text -> the cleartext oracle gives the netlist's own outputs."""
    from oracle import lut_oracle
    for case in ("full_adder", "adder8", "aes_sbox", "mul4"):
        rec = NETLISTS[case]
        env = parse_blif(rec["blif"])
        lut = map_basic(env)
        text = _printed(lut, show_outputs=True)
        ins = _inputs(rec)
        got = lut_oracle.eval_fbs_text(text, ins)
        _same_outputs(got, rec)


def test_command_line_loader_picks_the_format(tmp_path):
    from tfhe_fbs_map_amd.__main__ import load
    rec = NETLISTS["full_adder"]
    p = tmp_path / "fa.blif"
    p.write_text(rec["blif"])
    env, bits, kind = load(str(p), "auto", None)
    assert kind == "blif" and bits is not None and _printed(env, show_outputs=True).count("Bootstrap") == env.stats()["nb_bootstrap"]
    q = tmp_path / "fa.fbs"
    q.write_text(rec["basic_fbs"])
    env2, bits2, kind2 = load(str(q), "auto", None)
    assert kind2 == "fbs" and bits2 is None and env2.stats()["nb_bootstrap"] == env.stats()["nb_bootstrap"]
    buf = io.StringIO()
    env.write_lbf(os=buf)
    r = tmp_path / "fa.lbf"
    r.write_text(buf.getvalue())
    env3, _, kind3 = load(str(r), "auto", None)
    assert kind3 == "lbf" and env3.stats()["nb_bootstrap"] == env.stats()["nb_bootstrap"]
