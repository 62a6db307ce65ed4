"""GLWE dimension k = 2 at N = 1024 (k_blind_rotate_pairs_k2: three waves per bootstrap, twelve per CU, products added into the
components' exchange buffers with LDS atomics): word for word against the oracle, and through the drop-in API."""
import numpy as np
import pytest

from oracle import lut_oracle, tfhe_oracle as orc
from tests.helpers import load_fixture, oracle_eval_program, subsample, toy_k2

pytestmark = pytest.mark.gpu



@pytest.fixture(scope="module")
def nat():
    from tfhe_fbs_map_amd import _native
    return _native


TABLES = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1], [0, 1, 2, 3, 2, 1, 0], [1, 1, 1, 0, 1, 0, 0, 1, 1, 1]]


def toy(**kw):
    from tfhe_fbs_map_amd import Params
    base = dict(n=16, log_n_poly=10, k=2, l_bsk=1, beta_bsk=21, t_ksk=8, gamma_ksk=2, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2)
    base.update(kw)
    return Params(**base)


@pytest.mark.parametrize("beta", [21, 17, 12])
@pytest.mark.parametrize("shape", [0, 3])
def test_ragged_batches_bit_exact(nat, beta, shape):
    """Batches that are not multiples of the four bootstraps of a workgroup, all three table modes and a multi-valued table, a
    trivial ciphertext (every step skipped: the other three bootstraps of its workgroup still meet their barriers) and maximal
    residues: every output word equal to the oracle's -- by the launcher's own choice (shape 0: one bootstrap on the twelve waves
    of a workgroup up to three per CU, four-bootstrap workgroups beyond) and with the three-waves-per-bootstrap kernels forced
    at every size (shape 3: one, two, four bootstraps per workgroup)."""
    prm = toy(beta_bsk=beta)
    ctx, o = nat.Context(prm, seed=4), orc.Oracle(prm, seed=4)
    ctx.tune(br_k2_shape=shape)
    tv = ctx.tvset(TABLES)
    for B in (1, 2, 3, 4, 5, 7, 21, 64, 301, 600, 900):
        msgs = np.arange(B) % 7
        ids = (np.arange(B) % 4).astype(np.uint32)
        msgs[ids == 1] = np.arange(B)[ids == 1] % 14
        msgs[ids == 3] = np.arange(B)[ids == 3] % 10
        cts = ctx.encrypt(msgs, 3 + B)
        if B > 2:
            cts[B - 1, :-1] = 0
            cts[B // 2, :] = orc.Q - 1
        ctx.profile(True)
        ctx.profile_read(reset=True)
        got = ctx.bootstrap_batch(tv, cts, ids)
        if shape == 3:
            want = "k_blind_rotate_pairs_k2<10,%d>" % (1 if B <= 256 else 2 if B <= 512 else 4)
        else:
            want = "k_blind_rotate_cu_k2" if B <= 768 else "k_blind_rotate_pairs_k2<10,4>"
        assert want in ctx.profile_kernels(), (want, ctx.profile_kernels())
        ref, _ = o.bootstrap_batch(cts, TABLES, ids)
        assert np.array_equal(got, ref), B
    ctx.close()


def test_odd_steps_and_real_size_bit_exact(nat):
    """The 128-bit set the selector returns for (15, 70) with k = 2 admitted: six ciphertexts at full n against the oracle, a
    full round decrypted."""
    from tfhe_fbs_map_amd.params import choose_params, margin_sigmas, security_bits
    prm = choose_params(15, 70, glwe_dims=(1, 2))
    assert prm.k == 2 and prm.N == 1024 and prm.bsk_group == 2 and prm.l_bsk == 1
    assert security_bits(prm) >= 127.9 and margin_sigmas(prm, 70) >= 6.0
    ctx, o = nat.Context(prm, seed=1), orc.Oracle(prm, seed=1)
    rng = np.random.default_rng(42)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
    tv = ctx.tvset(tables)
    msgs = rng.integers(0, 15, 6)
    ids = (np.arange(6) % 16).astype(np.uint32)
    cts = ctx.encrypt(msgs, nonce0=5)
    ref, _ = o.bootstrap_batch(cts, tables, ids)
    assert np.array_equal(ctx.bootstrap_batch(tv, cts, ids), ref)
    B = 1024 + 37
    msgs = rng.integers(0, 15, B)
    ids = (np.arange(B) % 16).astype(np.uint32)
    out = ctx.bootstrap_batch(tv, ctx.encrypt(msgs, nonce0=100), ids)
    assert np.array_equal(ctx.decrypt(out), [tables[i][m] for i, m in zip(ids, msgs)])
    ctx.close()


@pytest.mark.parametrize("p, norm2", [(15, 70), (4, 2)])
def test_shipped_k2_sets_at_full_size_against_the_oracle(nat, p, norm2):
    """What `LutExecEnv.eval` runs by default for wide programs, as the bench runs it: the selector's own k = 2 set (p = 15 at
    norm2 70: n = 734; p = 4: n = 630) at full n, 1 024 ciphertexts through fbs_bootstrap_batch_dev on the throughput shape
    k_blind_rotate_pairs_k2<10,4> -- word for word against the oracle on ciphertexts that sit in every sub-slot (bootstrap
    0 .. 3 of a workgroup = three waves on other SIMDs each) of the first, a middle and the last workgroup, trivial
    ciphertexts (every step skipped: the bootstrap only keeps its workgroup's barriers company) beside ordinary ones.  Then a
    launch of 1 024 + 100, which the launcher CUTS -- a whole round, and the leftovers on the twelve-waves-per-bootstrap shape
    k_blind_rotate_cu_k2 -- checked at both ends and either side of the cut; then 300: two rounds of that shape alone, the
    second partial, checked at the ends of both."""
    import torch
    from tfhe_fbs_map_amd.params import choose_params, margin_sigmas, security_bits
    prm = choose_params(p, norm2, glwe_dims=(1, 2))
    assert prm.k == 2 and prm.N == 1024 and prm.bsk_group == 2 and prm.l_bsk == 1, "the selector moved: pin this test's parameter set"
    assert security_bits(prm) >= 127.9 and margin_sigmas(prm, norm2) >= 6.0
    ctx, o = nat.Context(prm, seed=1), orc.Oracle(prm, seed=1)
    rng = np.random.default_rng(7 + p)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    tv = ctx.tvset(tables)
    for B, pick, want in ((1024, [0, 1, 2, 3, 4, 5, 6, 7, 508, 509, 510, 511, 1020, 1021, 1022, 1023, 301, 778], ["k_blind_rotate_pairs_k2<10,4>"]),
                          (1124, [0, 3, 1021, 1023, 1024, 1025, 1026, 1027, 1100, 1120, 1121, 1122, 1123],
                           ["k_blind_rotate_cu_k2", "k_blind_rotate_pairs_k2<10,4>"]),
                          (300, [0, 1, 2, 254, 255, 256, 257, 298, 299], ["k_blind_rotate_cu_k2"])):
        msgs = rng.integers(0, p, B)
        ids = (np.arange(B) % 16).astype(np.uint32)
        cts = ctx.encrypt(msgs, nonce0=100)
        trivial = [i for i in (1, 6, 509, 1022, B - 1) if i < B]
        for i in trivial:
            cts[i, :-1] = 0                                          # mask zero: every modulus-switched mask word is zero
        d_in = torch.from_numpy(cts.view(np.int64)).cuda()
        d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
        d_out = torch.empty_like(d_in)
        ctx.profile(True)
        ctx.profile_read(reset=True)
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
        ctx.sync()
        launched = [k for k in ctx.profile_kernels() if "blind_rotate" in k]
        ctx.profile(False)
        assert sorted(launched) == want, launched
        got = d_out.cpu().numpy().view(np.uint64)
        ref, _ = o.bootstrap_batch(cts[pick], tables, ids[pick])
        assert np.array_equal(got[pick], ref), (B, launched)
        keep = np.ones(B, bool)
        keep[trivial] = False
        assert np.array_equal(ctx.decrypt(got)[keep], np.array([tables[i][m] for i, m in zip(ids, msgs)])[keep])
    ctx.close()


@pytest.mark.parametrize("name,T", [("demo_fbs_exec_env", 2), ("edge_outputs", 5), ("adder8__search_p7", 3), ("aes_sbox__search_p7", 2),
                                    ("mul16__search_p15", 2)])          # BASELINE config 3's program, every ciphertext
def test_program_ciphertexts_bit_exact_at_k2(nat, name, T):
    """Whole programs on a context of GLWE dimension 2 (fbs_eval: k_lincomb on ciphertexts of 2 N + 1 words, the shared key
    switch, k_blind_rotate_pairs_k2, wire slots) against the oracle evaluating the same instruction list one ciphertext at a
    time: every output word identical -- tests/test_gpu_parity.py::test_program_ciphertexts_bit_exact at k = 2."""
    from tfhe_fbs_map_amd import parse_fbs
    rec = load_fixture(name)
    ops, outs = lut_oracle.read_fbs(rec["fbs"])
    p = max(7, max(len(op[3]) for op in ops if op[0] == "boot"))
    prm = toy_k2(p)
    ctx, o = nat.Context(prm, seed=6), orc.Oracle(prm, seed=6)
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    ins, expect = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=9)
    assert cts.shape[-1] == 2 * 1024 + 1
    prog = nat.Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                       low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"])
    got = prog.eval(cts, T)
    wires = oracle_eval_program(o, ops, outs, {n: cts[i] for i, n in enumerate(low["input_names"])})
    for j, (out_name, src) in enumerate(outs):
        if src in ("0", "1"):
            assert ctx.decrypt(got[j]).tolist() == [int(src)] * T
        else:
            assert np.array_equal(got[j], wires[src]), out_name
            assert np.array_equal(ctx.decrypt(got[j]), expect[out_name])
    ctx.close()


def test_other_k2_shapes_take_the_general_kernel_or_are_refused(nat):
    """k = 2 at N = 1024 with one key bit per step or two levels, k = 3: k_blind_rotate_glwe (tests/test_gpu_glwe.py); N = 2048 at
    k = 2: no kernel, an error code at context creation."""
    from tfhe_fbs_map_amd import FbsError
    with pytest.raises(FbsError):
        nat.Context(toy(log_n_poly=11), seed=1)
    for kw in (dict(bsk_group=1), dict(l_bsk=2, beta_bsk=10), dict(k=3)):
        ctx, o = nat.Context(toy(**kw), seed=1), orc.Oracle(toy(**kw), seed=1)
        cts = ctx.encrypt(np.arange(5) % 7, 3)
        ids = (np.arange(5) % 4).astype(np.uint32)
        ctx.profile(True)
        got = ctx.bootstrap_batch(ctx.tvset(TABLES), cts, ids)
        assert any(k.startswith("k_blind_rotate_glwe<10,") and k.endswith(",1>") for k in ctx.profile_kernels())
        assert np.array_equal(got, o.bootstrap_batch(cts, TABLES, ids)[0])
        ctx.close()


def test_eval_takes_k2_at_every_width():
    """`LutExecEnv.eval` by default: the k = 2 set whether the program's levels are launches of a round of bootstraps (T = 200: the
    four-per-workgroup shape) or of a handful (T = 8: one bootstrap on the twelve waves of a workgroup) -- both decrypt to the
    reference's goldens; `glwe_dims=(1,)` keeps a program on the k = 1 set."""
    from tfhe_fbs_map_amd import ExecConfig, parse_fbs
    rec = load_fixture("mul16__search_p15")
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    for T, kernel in ((200, "k_blind_rotate_pairs_k2<10,4>"), (8, "k_blind_rotate_cu_k2")):
        cfg = ExecConfig(seed=9)
        ins, expect = subsample(rec, T)
        ctx, _ = cfg.choose(env, 15, samples=T)
        ctx.profile(True)
        ctx.profile_read(reset=True)
        got = env.eval(ins, config=cfg)
        assert cfg.last_choice["params"].k == 2 and cfg.last_choice["samples"] == T, (T, cfg.last_choice)
        assert kernel in ctx.profile_kernels(), (T, sorted(ctx.profile_kernels()))
        ctx.profile(False)
        for name, v in expect.items():
            assert (int(got[name]) == int(v)) if isinstance(v, int) else np.array_equal(np.asarray(got[name]).reshape(-1), v), name
    cfg = ExecConfig(seed=9, glwe_dims=(1,))
    ins, _ = subsample(rec, 200)
    env.eval(ins, config=cfg)
    assert cfg.last_choice["params"].k == 1


def test_noise_model_holds_at_k2(nat):
    """The variance model the selector rests on, at the k = 2 set for (15, 70): the measured bootstrap OUTPUT noise (the blind
    rotation's term: k + 1 = 3 key polynomials' noise per step, rounding seen through a key of k N = 2048 bits) against
    params.variances, and every output far inside its box.  The modulus switch onto 2N = 2048 slots - four times the variance of
    the N = 2048 sets', the term that sets n - is checked by chaining: bootstraps of bootstrap outputs scaled to norm2 70 decrypt."""
    from tfhe_fbs_map_amd.params import choose_params, variances
    prm = choose_params(15, 70, glwe_dims=(1, 2))
    assert prm.k == 2
    ctx, o = nat.Context(prm, seed=13), orc.Oracle(prm, seed=13)
    rng = np.random.default_rng(5)
    table = [0] + [int(v) for v in rng.integers(0, 2, 14)]
    B = 400
    msgs = rng.integers(0, 15, B)
    cts = ctx.encrypt(msgs, nonce0=900)
    tv = ctx.tvset([table])
    out = ctx.bootstrap_batch(tv, cts)
    assert np.array_equal(ctx.decrypt(out), [table[m] for m in msgs])
    phase = o.phase(out).astype(object)
    want = np.array([table[m] for m in msgs], dtype=object) * (2 * o.delta_half)
    err = np.array([min((int(p) - int(w)) % orc.Q, (int(w) - int(p)) % orc.Q) for p, w in zip(phase, want)], dtype=np.float64)
    predicted = np.sqrt(variances(prm)[0]) * orc.Q
    measured = float(np.sqrt(np.mean(err ** 2)))
    assert 0.5 * predicted < measured < 1.25 * predicted, (measured, predicted)
    assert err.max() < 0.1 * orc.Q / (4 * 15)
    # a linear combination of squared norm 70 of bootstrap outputs (8 x b0 + 2 x b1 + b2 + b3: 64 + 4 + 1 + 1), bootstrapped again
    bits = np.array([table[m] for m in msgs])
    idx = rng.integers(0, B, (4, B))
    coefs = np.array([8, 2, 1, 1])
    value = (coefs[:, None] * bits[idx]).sum(0)                      # 0 .. 12 < 15
    lc = np.zeros_like(out)
    for c, row in zip(coefs, idx):
        lc = (lc + int(c) * out[row].astype(object)) % orc.Q
    lc = lc.astype(np.uint64)
    ident = list(range(15))
    again = ctx.bootstrap_batch(ctx.tvset([ident]), lc)
    assert np.array_equal(ctx.decrypt(again), value)
    ctx.close()
