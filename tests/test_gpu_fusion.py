"""Several tables on ONE blind rotation (include/fbs_exec.h FBS_LOAD_FUSE_TABLES).  The reference's
one-gate-one-bootstrap lowering puts several tables on one linear combination (map_to_fbs.py:41-45; its CSE merges
identical tables only, fbs_exec_env.py:93-100).  A fused program rotates such a source once, from the table-independent
TV_0, and cuts every table out of that accumulator (TV_F = TV_0 * D_F).  Checked word for word against the CPU oracle's
restatement of the same construction, and at the decrypted level against the reference's cleartext goldens."""
import numpy as np
import pytest

from oracle import lut_oracle, tfhe_oracle as orc
from tests.helpers import load_fixture, oracle_eval_program, subsample, toy_k2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from tfhe_fbs_map_amd import _native
    return _native


def load(nat, prm, text, inputs, seed=6, fuse=True, merge=True):
    from tfhe_fbs_map_amd import parse_fbs
    ctx = nat.Context(prm, seed=seed)
    env = parse_fbs(text, inputs=inputs, merge_linear_prods=merge)
    low = env.lower()
    tv = ctx.tvset(low["tables"])
    prog = nat.Program(ctx, tv, len(low["input_names"]), low["kind"], low["arg0"], low["arg1"], low["const_coef"],
                       low["term_coef"], low["term_src"], low["out_wire"], fuse_tables=fuse)
    return ctx, low, tv, prog


def params_for(toy_params, rec, k=1):
    ops, _ = lut_oracle.read_fbs(rec["fbs"])
    p = max(7, max(len(op[3]) for op in ops if op[0] == "boot"))
    return toy_params.replace(p_msg=p) if k == 1 else toy_k2(p)      # k = 2: accumulator rows of 3 N words, ciphertexts of 2 N + 1


@pytest.mark.parametrize("name,k", [("adder8__basic_p2", 1), ("2_input_gates__basic_p2", 1), ("half_adder__basic_p2", 1),
                                    ("adder8__basic_p2", 2), ("2_input_gates__basic_p2", 2)])       # round 4: shared rotations at GLWE dimension 2
def test_fused_program_equals_the_oracle_and_the_reference(nat, toy_params, name, k):
    T = 3
    rec = load_fixture(name)
    prm = params_for(toy_params, rec, k)
    ctx, low, tv, prog = load(nat, prm, rec["fbs"], rec["program_inputs"])
    ops, outs = lut_oracle.read_fbs(rec["fbs"])
    distinct = len({op[2] for op in ops if op[0] == "boot"})
    assert prog.n_rotations == distinct < prog.n_bootstrap          # one rotation per source, however many tables read it
    assert prog.n_keyswitch == distinct
    ins, expect = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=9)
    got = prog.eval(cts, T)
    o = orc.Oracle(prm, seed=6)
    in_cts = {n: cts[i] for i, n in enumerate(low["input_names"])}
    fused = oracle_eval_program(o, ops, outs, in_cts, fuse=True)
    plain = oracle_eval_program(o, ops, outs, in_cts, fuse=False)
    differs = False
    for k, (out_name, src) in enumerate(outs):
        if src in ("0", "1"):
            continue
        assert np.array_equal(got[k], fused[src]), out_name
        assert np.array_equal(ctx.decrypt(got[k]), expect[out_name])
        differs |= not np.array_equal(fused[src], plain[src])
    assert differs                                                    # other ciphertexts than the unfused program's, same plaintexts


def test_a_program_with_nothing_to_share_is_unchanged(nat, toy_params):
    rec = load_fixture("aes_sbox__basic_p2")
    prm = params_for(toy_params, rec)
    ctx, low, tv, fused = load(nat, prm, rec["fbs"], rec["program_inputs"])
    assert fused.n_rotations == fused.n_bootstrap
    plain = nat.Program(ctx, tv, len(low["input_names"]), low["kind"], low["arg0"], low["arg1"], low["const_coef"],
                        low["term_coef"], low["term_src"], low["out_wire"])
    ins, _ = subsample(rec, 2)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=3)
    assert np.array_equal(fused.eval(cts, 2), plain.eval(cts, 2))


MULTI = """m1 = 1 * a + 4 * b + 8 * c
m2 = Bootstrap(m1, [0, 1, 2, 1, 0, 2, 1, 2, 1, 0, 1, 2, 0, 1])
m3 = Bootstrap(m1, [1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0])
m4 = Bootstrap(m1, [0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 0, 1, 0, 0])
m5 = 1 * a + 1 * b
m6 = Bootstrap(m5, [0, 1, 1])
m7 = 1 * m6 + 2 * m3 + 3 * a
m8 = Bootstrap(m7, [0, 1, 1, 0, 1, 0, 0])
m9 = Bootstrap(m7, [0, 0, 0, 1, 0, 1, 1])
m10 = Bootstrap(m7, [1, 1, 0, 0, 0, 0, 1])
Output o1 = m2
Output o2 = m8
Output o3 = m9
Output o4 = m10
Output o5 = m6
Output o6 = m4
"""


@pytest.mark.parametrize("log_n,group", [(10, 1), (11, 1), (11, 2), (8, 1)])
def test_multi_valued_and_negacyclic_tables_share_a_rotation(nat, log_n, group):
    """Tables with values beyond {0, 1}, negacyclic tables (c = 1, 2), a source with one reader beside shared ones; at N = 2048 the accumulator is written by four waves, with two key bits per step by the pairs kernel."""
    from tfhe_fbs_map_amd import Params
    prm = Params(n=12, log_n_poly=log_n, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=1 << 8, bsk_group=group)
    T = 8
    ctx, low, tv, prog = load(nat, prm, MULTI, ["a", "b", "c"])
    assert (prog.n_bootstrap, prog.n_rotations, prog.n_keyswitch) == (7, 3, 3)
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, (3, T))
    cts = ctx.encrypt(bits, nonce0=2)
    got = prog.eval(cts, T)
    ops, outs = lut_oracle.read_fbs(MULTI)
    o = orc.Oracle(prm, seed=6)
    wires = oracle_eval_program(o, ops, outs, {n: cts[i] for i, n in enumerate(low["input_names"])}, fuse=True)
    clear = lut_oracle.eval_fbs_text(MULTI, {n: bits[i] for i, n in enumerate(low["input_names"])})
    for k, (out_name, src) in enumerate(outs):
        assert np.array_equal(got[k], wires[src]), out_name
        assert np.array_equal(ctx.decrypt(got[k]) % (2 * prm.p_msg), np.asarray(clear[out_name]) % (2 * prm.p_msg)), out_name
    # what the library reports about the noise of shared outputs, against the oracle's own polynomials and the facade's closed forms
    from tfhe_fbs_map_amd.fbs_exec_env import table_fusion_norms
    for t, table in enumerate(low["tables"]):
        d, _ = o.build_tv_diff(table)
        tvp, _ = o.build_tv(table)
        g = np.where(tvp > orc.Q // 2, tvp.astype(np.int64) - orc.Q, tvp.astype(np.int64)) // o.delta_half
        assert tv.fusion_norms(t) == (int((d.astype(np.int64) ** 2).sum()), int((g ** 2).sum()))
        assert table_fusion_norms(table, prm.p_msg)[0] == tv.fusion_norms(t)[0]


def test_tables_with_negative_values_below_the_facade(nat, toy_params):
    """c = 0 tables (f(x + p) = -f(x)) and negative entries: not expressible through the builder (its tables start at
    0, fbs_exec_env.py:76), legal at the C ABI."""
    prm = toy_params
    tables = [[0, 1, 2, 3, 4, 5, 6, 2, 1, 0, -1, -2, -3, -4], [0, 0, 0, 1, 0, 1, 1, 0, 0, 0, -1, 0, -1, -1], [3, -2, 5, 0, 0, -6, 1]]
    ctx = nat.Context(prm, seed=6)
    tv = ctx.tvset(tables)
    #       m1 = a + 4 b + 8 c           three tables on m1
    prog = nat.Program(ctx, tv, 3, [0, 1, 1, 1], [0, 3, 3, 3], [3, 0, 1, 2], [0, 0, 0, 0], [1, 4, 8], [0, 1, 2], [4, 5, 6],
                       fuse_tables=True)
    assert (prog.n_bootstrap, prog.n_rotations) == (3, 1)
    T = 8
    bits = np.array([[0, 1, 0, 1, 0, 1, 0, 1], [0, 0, 1, 1, 0, 0, 1, 1], [0, 0, 0, 0, 1, 1, 1, 1]])
    cts = ctx.encrypt(bits, nonce0=4)
    got = prog.eval(cts, T)
    o = orc.Oracle(prm, seed=6)
    ops = [("lin", "m1", [(1, "a"), (4, "b"), (8, "c")], 0)] + [("boot", "t%d" % i, "m1", t) for i, t in enumerate(tables)]
    wires = oracle_eval_program(o, ops, [], dict(a=cts[0], b=cts[1], c=cts[2]), fuse=True)
    v = bits[0] + 4 * bits[1] + 8 * bits[2]
    for i, t in enumerate(tables):
        assert np.array_equal(got[i], wires["t%d" % i])
        full = t if len(t) == 14 else t + [-x for x in t]          # a table that stops at p continues negacyclically
        assert np.array_equal(ctx.decrypt(got[i]) % 14, np.array([full[x] for x in v]) % 14)


@pytest.mark.parametrize("k", [1, 2])
def test_levels_of_a_fused_program_whole_or_sliced_into_rows(nat, toy_params, k):
    """Into the wire slots a level with shared rotations runs whole (a partial range is refused).  Into ROWS it can be cut
    anywhere: rows are `row_words` = (k + 1) N words, a shared rotation leaves its accumulator in its row, and the scatter call cuts
    the tables out of the gathered rows -- same ciphertexts as the whole evaluation."""
    import torch
    rec = load_fixture("adder8__basic_p2")
    prm = params_for(toy_params, rec, k)
    T = 4
    ctx, low, tv, prog = load(nat, prm, rec["fbs"], rec["program_inputs"])
    assert prog.row_words == (k + 1) * prm.N
    ins, _ = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=9)
    ref = prog.eval(cts, T)
    ctw = prm.ct_words
    wires = torch.zeros((prog.n_slots, T, ctw), dtype=torch.int64, device="cuda")
    wires[torch.from_numpy(prog.in_slot.astype(np.int64)).cuda()] = torch.from_numpy(cts.view(np.int64)).cuda()
    refused = 0
    for L in range(prog.depth + 1):
        prog.level_lincomb_dev(L, wires.data_ptr(), T, 0, T)
        if L == prog.depth:
            break
        total = prog.level_width[L] * T
        if total > 1:
            try:                                               # part of a level straight into the slots: refused where tables share a rotation
                prog.level_bootstrap_dev(L, wires.data_ptr(), T, 0, T, 0, total // 2)
            except nat.FbsError as e:
                assert "runs whole" in str(e)
                refused += 1
        rows = torch.empty((total, prog.row_words), dtype=torch.int64, device="cuda")
        cut = total // 3                                       # two unequal slices, as two ranks would compute them
        for f0, f1 in ((0, cut), (cut, total)):
            if f1 > f0:
                prog.level_bootstrap_dev(L, wires.data_ptr(), T, 0, T, f0, f1, d_rows=rows[f0:].data_ptr())
        prog.level_scatter_dev(L, wires.data_ptr(), T, 0, T, rows.data_ptr(), 0, total)
    ctx.sync()
    assert refused > 0
    got = wires.cpu().numpy().view(np.uint64)
    for k, slot in enumerate(prog.out_slot.tolist()):
        if slot >= 0:
            assert np.array_equal(got[slot], ref[k])


def test_fused_samples_in_chunks(nat, toy_params, monkeypatch):
    rec = load_fixture("adder8__basic_p2")
    prm = params_for(toy_params, rec)
    T = 29
    ctx, low, tv, prog = load(nat, prm, rec["fbs"], rec["program_inputs"])
    ins, expect = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=9)
    ref = prog.eval(cts, T)
    monkeypatch.setenv("FBS_WIRE_BUDGET_MB", "2")
    ctx2, low2, tv2, prog2 = load(nat, prm, rec["fbs"], rec["program_inputs"])
    assert np.array_equal(prog2.eval(cts, T), ref)
    for k, name in enumerate(low["out_names"]):
        assert np.array_equal(ctx2.decrypt(ref[k]), expect[name])


def test_noise_of_a_shared_rotation_against_the_model(nat):
    """Output noise variance of tables cut out of a shared rotation against an ordinary bootstrap's, measured on 4096
    bootstraps per table at the reduced-noise benchmark set, where the rounding of the accumulator is the whole of the
    rotation noise.  Through the binary GLWE key S that part is seen as |D_F|^2 + |S D_F|^2 against 1 + |S|^2 -- about
    |D_F|^2 / 2 + mean(G_F^2) / 2 because S has mean 1/2 and (1 + .. + X^(N-1)) D_F = G_F.  `table_fusion_factor`, which
    the parameter choice of a fused evaluation carries, bounds it."""
    from tfhe_fbs_map_amd import P1024
    from tfhe_fbs_map_amd.fbs_exec_env import table_fusion_factor, table_fusion_norms
    prm = P1024.replace(p_msg=7, n=64)                 # (short rotation: the ratio does not depend on n)
    tabs = [[0, 1, 1, 0], [0, 1, 0, 1], [0, 1, 2, 3]]
    text = "m1 = 1 * a + 2 * b\n" + "".join("m%d = Bootstrap(m1, %s)\n" % (k + 2, t) for k, t in enumerate(tabs)) + \
        "".join("Output o%d = m%d\n" % (k, k + 2) for k in range(len(tabs)))
    T = 4096
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2, (2, T))
    v = bits[0] + 2 * bits[1]
    o = orc.Oracle(prm, seed=6)
    delta = 2 * o.delta_half
    var = {}
    for fuse in (False, True):
        ctx, low, tv, prog = load(nat, prm, text, ["a", "b"], fuse=fuse)
        assert prog.n_rotations == (1 if fuse else 3)
        out = prog.eval(ctx.encrypt(bits, nonce0=1), T)
        msgs = ctx.decrypt(out)
        for k, t in enumerate(tabs):
            assert np.array_equal(msgs[k], np.array(t)[v])
        err = (o.phase(out).astype(np.int64) - msgs * delta + orc.Q // 2) % orc.Q - orc.Q // 2
        var[fuse] = (err.astype(np.float64) ** 2).mean(axis=1)
    S = o.keys()["sk_glwe"].astype(np.int64)                     # GLWE key, k = 1
    N = prm.N
    for k, t in enumerate(tabs):
        d, _ = o.build_tv_diff(t)
        sd = np.zeros(N, np.int64)                         # S * D_F mod X^N + 1
        for i in np.nonzero(d)[0]:
            sd += int(d[i]) * np.concatenate([-S[N - i:], S[:N - i]])
        d2 = int((d.astype(np.int64) ** 2).sum())
        predicted = (d2 + float((sd ** 2).sum())) / (1.0 + float((S ** 2).sum()))
        d2_closed, g2_mean = table_fusion_norms(t, prm.p_msg)
        assert d2_closed == d2 and abs(predicted - (d2 / 2 + g2_mean / 2)) < 0.12 * predicted, (t, predicted)
        ratio = var[True][k] / var[False][k]
        assert 0.85 * predicted < ratio < 1.15 * predicted, (t, ratio, predicted)
        assert ratio < 1.05 * table_fusion_factor(t, prm.p_msg), (t, ratio)


BASIC = ["adder8__basic_p2", "2_input_gates__basic_p2", "half_adder__basic_p2", "full_adder__basic_p2", "aes_sbox__basic_p2", "trivium_iter_v2__basic_p2"]


@pytest.mark.parametrize("name", BASIC)
def test_default_config_fuses_where_it_pays_and_decrypts_to_the_reference(name):
    """`LutExecEnv.eval` with the default ExecConfig: 128-bit parameters chosen for the FUSED noise statistic, shared
    rotations where the program has any; every output of every harness sample equals the reference's cleartext."""
    from tfhe_fbs_map_amd import ExecConfig, parse_fbs, security_bits
    from tfhe_fbs_map_amd.params import margin_sigmas
    try:
        rec = load_fixture(name)
    except FileNotFoundError:
        pytest.skip("no such fixture")
    T = 200
    ins, expect = subsample(rec, T)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    cfg = ExecConfig(seed=5)
    got = env.eval(ins, config=cfg)
    for k, v in expect.items():
        assert np.array_equal(np.asarray(got[k]).reshape(-1), np.asarray(v).reshape(-1)), k
    (prog, _), = cfg._programs.values()
    stats = env.stats()
    p = prog.ctx.params.p_msg
    fstats = env.fusion_stats(p)
    assert prog.fused == (fstats["nb_rotation"] < stats["nb_bootstrap"])
    assert prog.n_rotations == fstats["nb_rotation"] and prog.n_bootstrap == stats["nb_bootstrap"]
    assert security_bits(prog.ctx.params) >= 127.9
    assert margin_sigmas(prog.ctx.params, fstats["norm2_linprod"] if prog.fused else stats["norm2_linprod"]) >= 5.9
    off = ExecConfig(seed=5, fuse_tables=False)                          # the switch: same plaintexts, every table its own rotation
    got2 = env.eval(ins, config=off)
    for k, v in expect.items():
        assert np.array_equal(np.asarray(got2[k]).reshape(-1), np.asarray(v).reshape(-1)), k
    (prog2, _), = off._programs.values()
    assert not prog2.fused and prog2.n_rotations == stats["nb_bootstrap"]


def test_load_options_and_queries_fail_with_codes(nat, toy_params):
    import ctypes as C
    ctx = nat.Context(toy_params, seed=6)
    tv = ctx.tvset([[0, 1, 1, 0]])
    keep = (np.array([1], np.uint8), np.array([0], np.uint32), np.array([0], np.uint32), np.array([0], np.int64),
            np.zeros(1, np.int64), np.zeros(1, np.uint32), np.array([1], np.int64))        # one Bootstrap of the input
    desc = nat._ProgramDesc(1, 1, 0, 1, *[a.ctypes.data for a in keep])
    h = C.c_void_p()
    assert nat.lib.fbs_program_load_ex(ctx._h, C.byref(desc), tv._h, 2, C.byref(h)) == -1 and not h.value     # unknown flag
    assert "unknown load flag" in nat.lib.fbs_last_error(ctx._h).decode()
    d, g = C.c_uint64(), C.c_uint64()
    assert nat.lib.fbs_table_fusion_norms(tv._h, 1, C.byref(d), C.byref(g)) == -1                               # no such table
    assert nat.lib.fbs_table_fusion_norms(tv._h, 0, C.byref(d), None) == 0 and d.value == 2
    with pytest.raises(nat.FbsError):
        tv.fusion_norms(7)


def test_shared_rotations_in_a_level_longer_than_a_round(nat):
    """The benchmark shape (l = 3, beta = 7, N = 1024) launches whole rounds as whole-CU workgroups and the rest separately
    (whole_cu_share): the accumulators of shared rotations must land in their scratch rows on both sides of the cut."""
    from tfhe_fbs_map_amd import P1024
    prm = P1024.replace(n=16, p_msg=7)
    T = 700                                                # level 1: 2 rotations x 700 = a round and 376
    ctx, low, tv, prog = load(nat, prm, MULTI, ["a", "b", "c"])
    assert prog.n_rotations == 3 and prog.n_bootstrap == 7
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2, (3, T))
    cts = ctx.encrypt(bits, nonce0=2)
    got = prog.eval(cts, T)
    clear = lut_oracle.eval_fbs_text(MULTI, {n: bits[i] for i, n in enumerate(low["input_names"])})
    for k, name in enumerate(low["out_names"]):
        assert np.array_equal(ctx.decrypt(got[k]) % 14, np.asarray(clear[name]) % 14), name
    ops, outs = lut_oracle.read_fbs(MULTI)                 # and word for word against the oracle on the samples around the cut
    o = orc.Oracle(prm, seed=6)
    pick = np.array([0, 1, 322, 323, 324, 325, 699])       # 1024 = 700 + 324: the cut falls at sample 324 of the second rotation
    wires = oracle_eval_program(o, ops, outs, {n: cts[i][pick] for i, n in enumerate(low["input_names"])}, fuse=True)
    for k, (out_name, src) in enumerate(outs):
        assert np.array_equal(got[k][pick], wires[src]), out_name
