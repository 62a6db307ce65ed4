"""The cleartext oracle (oracle/lut_oracle.py) against everything the reference pins for this path:
the golden fixtures captured by import (tests/golden/capture_reference.py) and the harness-input KAT."""
import numpy as np
import pytest

from oracle import lut_oracle
from tests.helpers import assert_outputs_equal, fixture_names, load_fixture

ALL = fixture_names()


def test_fixture_inventory():
    assert len(ALL) >= 200
    for must in ("demo_fbs_exec_env", "edge_outputs", "edge_nomerge", "adder128__search_p15", "adder128__search_p31",
                 "mul16__search_p15", "trivium_stream_short128__search_p15"):
        assert must in ALL


def test_harness_inputs_kat():
    # reference fbs_mapper/map_circuit.py:137-139 -- seed 42, one randint(0,2,1000) per input
    np.random.seed(42)
    a = np.random.randint(0, 2, (1000))
    b = np.random.randint(0, 2, (1000))
    assert "".join(map(str, a[:32])) == "01000100010000101110101111111100" and a.sum() == 510
    assert "".join(map(str, b[:16])) == "1000001011001001" and b.sum() == 474
    rec = load_fixture("half_adder__search_p3")
    assert np.array_equal(rec["inputs"][rec["harness_inputs"][0]], a)
    assert np.array_equal(rec["inputs"][rec["harness_inputs"][1]], b)


def test_demo_known_answers():
    # reference's __main__ demo, fbs_exec_env.py:279-301
    rec = load_fixture("demo_fbs_exec_env")
    out = lut_oracle.eval_fbs_text(rec["fbs"], rec["inputs"])
    assert {k: list(v) for k, v in out.items()} == {"f": [0, 0], "g": [2, 0], "h": [0, 1]}
    # map_to_fbs.py:550-596 demo
    for m in ("basic", "naive", "search"):
        rec = load_fixture("demo_map_to_fbs__" + m)
        out = lut_oracle.eval_fbs_text(rec["fbs"], rec["inputs"])
        assert {k: list(v) for k, v in out.items()} == {"d": [0, 0, 0, 1], "e": [0, 0, 1, 0], "f": [0, 0, 0, 1]}


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_reference_eval(name):
    rec = load_fixture(name)
    got = lut_oracle.eval_fbs_text(rec["fbs"], rec["inputs"])
    assert_outputs_equal(got, rec["outputs"])
    if rec.get("outputs_bitenv"):          # the reference's own self-check, map_circuit.py:174-180
        assert_outputs_equal(got, rec["outputs_bitenv"])


def test_wire_values_stay_in_table_range():
    for name in ("mul16__search_p15", "adder128__search_p31", "aes_sbox__search_p7"):
        rec = load_fixture(name)
        ops, outs = lut_oracle.read_fbs(rec["fbs"])
        wires = lut_oracle.evaluate(ops, outs, rec["inputs"], all_wires=True)
        for op in ops:
            if op[0] == "boot":
                v = wires[op[2]]
                assert v.min() >= 0 and v.max() < len(op[3])
                assert len(op[3]) == rec["max_val"][op[2]] + 1
