"""Host-side mirror of the reference's LutExecEnv (fbs_mapper/fbs_exec_env.py): builder, CSE, bounds,
serialisers, readers, scheduler and lowering -- all CPU, checked against the captured fixtures."""
import io

import numpy as np
import pytest

from oracle import lut_oracle
from tests.helpers import assert_outputs_equal, fixture_names, load_fixture
from tfhe_fbs_map_amd.fbs_exec_env import (FbsExecEnv, LutExecEnv, min_fbs_size, parse_fbs, parse_lbf,
                                          table_is_valid)

ALL = fixture_names()


def text_of(env, **kw):
    buf = io.StringIO()
    env.print(os=buf, **kw)
    return buf.getvalue()


def lbf_of(env):
    buf = io.StringIO()
    env.write_lbf(os=buf)
    return buf.getvalue()


def test_alias():
    assert FbsExecEnv is LutExecEnv


def test_builder_reproduces_reference_demo():
    # same calls as the reference's __main__ (fbs_exec_env.py:279-301) -> same program text
    env = LutExecEnv()
    a, b, c = env.input("a"), env.input("b"), env.const(1)
    d = env.linear([1, 2], [a, b])
    e = env.linear([1, 1], [c, d])
    f = env.bootstrap(e, [1, 0, 1, 1, 0])
    g = env.linear([2, 1], [a, f])
    h = env.bootstrap(g, [1, 1, 0, 2])
    env.bootstrap(h, [1, 0, 1])
    env.output("f", f); env.output("g", g); env.output("h", h)
    rec = load_fixture("demo_fbs_exec_env")
    assert text_of(env, show_outputs=True) == rec["fbs"]
    assert env.stats() == rec["stats"]
    assert {k: int(v) for k, v in env.max_val.items()} == rec["max_val"]
    assert lbf_of(env) == rec["lbf"]


def test_builder_edge_cases_match_reference():
    env = LutExecEnv()
    a, b = env.input("a"), env.input("b")
    na = env.linear([-1], [a], const_coef=1)
    s = env.linear([1, 1], [a, b])
    x1 = env.bootstrap(s, [0, 1, 0])
    x2 = env.bootstrap(s, [0, 1, 0])           # CSE: same node back, but the id is consumed
    assert x2 is x1
    c1 = env.bootstrap(s, [0, 0, 1])
    t = env.linear([1, 2, 1], [x1, c1, env.const(1)])
    y = env.bootstrap(t, [0, 1, 2, 3, 2])
    env.output("pa", a); env.output("z", env.const(0)); env.output("one", env.const(1))
    env.output("na", na); env.output("x", x2); env.output("y", y)
    rec = load_fixture("edge_outputs")
    assert text_of(env, show_outputs=True) == rec["fbs"]
    assert lbf_of(env) == rec["lbf"]
    assert env.stats() == rec["stats"]

    env = LutExecEnv(merge_linear_prods=False)
    a, b, c = env.input("a"), env.input("b"), env.input("c")
    l1 = env.linear([1, 1], [a, b])
    l2 = env.linear([2, 1], [l1, c], const_coef=1)
    z = env.bootstrap(l2, [0, 1, 1, 0, 1, 0, 1])
    env.output("z", z); env.output("l2", l2)
    rec = load_fixture("edge_nomerge")
    assert text_of(env, show_outputs=True) == rec["fbs"]
    assert env.stats() == rec["stats"]


def test_builder_assertions():
    env = LutExecEnv()
    a = env.input("a")
    with pytest.raises(AssertionError):
        env.bootstrap(a, [0, 1, 1])             # table length must be max_val + 1 (reference :150)
    with pytest.raises(AssertionError):
        env.bootstrap(a, (0, 1))                # must be a list (:149)
    with pytest.raises(AssertionError):
        env.linear([1], ["a"])                  # must be a Node (:134)
    with pytest.raises(AssertionError):
        env.output("o", "a")
    with pytest.raises(AssertionError):
        env.bootstrap(a, [1, 2])                # tables start at 0 (:86)
    assert env.const(3).name == "3" and str(env.const(3)) == "3"
    assert str(a) == "Input(a)" and a == env.input("a")


def test_show_inputs_and_dangling():
    env = LutExecEnv()
    a, b = env.input("a"), env.input("b")
    s = env.linear([1, 1], [a, b])
    keep = env.bootstrap(s, [0, 1, 0])
    env.bootstrap(s, [0, 0, 1])                 # dead
    env.output("o", keep)
    assert text_of(env, show_inputs=True).startswith("a = Input(a)\nb = Input(b)\n")
    env.remove_dangling_nodes()
    assert [i.name for i in env.instructions] == ["a", "b", "m1", "m2"]


@pytest.mark.parametrize("name", ALL)
def test_fbs_reader_roundtrip(name):
    rec = load_fixture(name)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"],
                    merge_linear_prods=name != "edge_nomerge")
    assert text_of(env, show_outputs=True) == rec["fbs"]
    assert env.stats() == rec["stats"]
    # (the reference keeps bounds of instructions that remove_dangling_nodes dropped; compare the live ones)
    assert all(rec["max_val"][k] == int(v) for k, v in env.max_val.items())
    if rec["lbf"] is not None:
        assert lbf_of(env) == rec["lbf"]


@pytest.mark.parametrize("name", [n for n in ALL if "search_p15" in n or n.startswith(("demo", "edge_outputs"))][:40])
def test_lbf_reader_is_semantically_equal(name):
    rec = load_fixture(name)
    if rec["lbf"] is None:
        pytest.skip("reference cannot write this program as .lbf")
    if any(v >= 10 for t in parse_fbs(rec["fbs"]).tables() for v in t):
        pytest.skip(".lbf table digits are ambiguous for entries >= 10 (SURVEY appendix B)")
    env = parse_lbf(rec["lbf"])
    assert env.stats() == rec["stats"]
    got = lut_oracle.eval_fbs_text(text_of(env, show_outputs=True), rec["inputs"])
    assert_outputs_equal(got, rec["outputs"])


def test_inputs_inferred_without_hint():
    rec = load_fixture("full_adder__search_p7")
    env = parse_fbs(rec["fbs"])
    assert sorted(i.name for i in env.instructions if isinstance(i, LutExecEnv.Input)) == sorted(rec["program_inputs"])


def test_table_contract():
    # reference map_to_fbs.py:81-98
    assert table_is_valid([0, 1, 1], 3) and table_is_valid([0, 1, 1], 7)
    assert table_is_valid([0, 1, 1, 1, 0, 0], 3)          # mode 1
    assert table_is_valid([0, 0, 1, 0, 0], 3)             # mode 2
    assert table_is_valid([1, 1, 0, 1, 1], 3)             # mode 3
    assert not table_is_valid([0, 1, 1, 1, 1, 0], 3)
    assert not table_is_valid([0] * 7, 3)
    assert min_fbs_size([[0, 1, 1, 0]]) == 2 and min_fbs_size([[0, 0, 0, 1]]) == 3 and min_fbs_size([[0, 1, 1, 0, 1, 0]]) == 4


@pytest.mark.parametrize("name", [n for n in ALL if "__naive_" in n or "__search_" in n])
def test_mapped_tables_are_evaluable_at_their_fbs_size(name):
    rec = load_fixture(name)
    p = rec["meta"]["fbs_size"]
    for t in parse_fbs(rec["fbs"]).tables():
        assert table_is_valid(t, p), (name, t)
        if rec["meta"]["strict"]:
            assert len(t) <= p


def test_schedule_and_lowering():
    rec = load_fixture("mul16__search_p15")
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    level, depth, widths = env.schedule()
    assert sum(widths) == rec["stats"]["nb_bootstrap"] and depth == len(widths) and min(widths) >= 1
    low = env.lower()
    n_in = len(low["input_names"])
    assert n_in == 32 and len(low["kind"]) == rec["stats"]["nb_linprod"] + rec["stats"]["nb_bootstrap"]
    # every instruction reads only earlier wires (what fbs_program_load requires)
    for i, k in enumerate(low["kind"]):
        w = n_in + i
        if k == 0:
            assert all(s < w for s in low["term_src"][low["arg0"][i]:low["arg0"][i] + low["arg1"][i]])
        else:
            assert low["arg0"][i] < w and low["arg1"][i] < len(low["tables"])
    assert len(low["tables"]) == len({tuple(t) for t in env.tables()})
    low2 = parse_fbs(load_fixture("edge_outputs")["fbs"]).lower()
    assert low2["out_wire"][low2["out_names"].index("z")] == -1 and low2["out_wire"][low2["out_names"].index("one")] == -2


def test_trivium_levels_are_wide():
    rec = load_fixture("trivium_stream_short128__search_p15")
    _, depth, widths = parse_fbs(rec["fbs"]).schedule()
    assert depth <= 8 and max(widths) >= 60


def test_exec_config_seeds_and_nonces():
    """No GPU needed: the key seed is 32 fresh bytes unless the caller makes it a constant, and encryption streams are the
    context's own counter (fbs_encrypt_fresh) unless the caller pins them."""
    from tfhe_fbs_map_amd import ExecConfig
    cfg = ExecConfig()
    assert cfg.take_nonces(12) is None                       # None -> Context.encrypt(nonce0=None) -> fbs_encrypt_fresh
    a, b = ExecConfig().key_seed(), ExecConfig().key_seed()
    assert isinstance(a, bytes) and len(a) == 32 and a != b and cfg.key_seed() == cfg.key_seed()
    fixed = ExecConfig(seed=7, nonce0=100)
    assert fixed.key_seed() == 7 and fixed.take_nonces(9) == 100 and fixed.take_nonces(9) == 100     # explicit = reproducible


def test_margin_floor_is_opt_in():
    """Where nothing reaches the asked margin the selector raises and says how to relax; it steps down only when told to."""
    import pytest
    from tfhe_fbs_map_amd import ExecConfig
    from tfhe_fbs_map_amd.params import margin_sigmas
    with pytest.raises(ValueError, match="allow_margin_floor"):
        ExecConfig(min_margin=40.0).params_choice(15, 70)
    prm = ExecConfig(min_margin=40.0, allow_margin_floor=4.0).params_choice(15, 70)
    assert 4.0 <= margin_sigmas(prm, 70) < 40.0


def test_fusion_statistics_of_a_one_gate_one_bootstrap_program():
    """What sharing blind rotations changes: the rotations left and the noise statistic the parameter
    choice has to carry.  adder8 under the reference's Basic lowering: XOR and AND of the same pair of wires are two
    tables on one linear combination (map_to_fbs.py:41-45)."""
    from tfhe_fbs_map_amd.fbs_exec_env import ExecConfig, table_fusion_factor, table_fusion_norms
    rec = load_fixture("adder8__basic_p2")
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    p = min_fbs_size(env.tables())
    stats, fused = env.stats(), env.fusion_stats(p)
    assert (stats["nb_bootstrap"], fused["nb_rotation"]) == (37, 22)
    assert p == 3 and table_fusion_norms([0, 1, 1, 0], p) == (2, 8 / 3) and table_fusion_norms([0, 0, 0, 1], p) == (1, 1)
    assert table_fusion_factor([0, 1, 1, 0], p) == pytest.approx(7 / 3) and table_fusion_factor([0, 0, 0, 1], p) == 1
    # 2 * xor + 1 * and of shared rotations: 4 * 7/3 + 1 * 1 against the reference's 4 + 1
    assert stats["norm2_linprod"] == 5 and fused["norm2_linprod"] == pytest.approx(4 * 7 / 3 + 1)
    # nothing shared: nothing changes
    rec = load_fixture("aes_sbox__basic_p2")
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    assert env.fusion_stats(3) == dict(nb_rotation=env.stats()["nb_bootstrap"], norm2_linprod=env.stats()["norm2_linprod"])
    # the parameter set for the fused statistic keeps the margin at THAT norm
    from tfhe_fbs_map_amd.params import margin_sigmas
    cfg = ExecConfig()
    assert margin_sigmas(cfg.params_choice(p, fused["norm2_linprod"]), fused["norm2_linprod"]) >= cfg.min_margin
