"""GLWE dimension k = 2 at N = 1024 (k_blind_rotate_pairs_k2: three waves per bootstrap, twelve per CU, products added into the
components' exchange buffers with LDS atomics): word for word against the oracle, and through the drop-in API."""
import numpy as np
import pytest

from oracle import tfhe_oracle as orc
from tests.helpers import load_fixture, subsample

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
def test_ragged_batches_bit_exact(nat, beta):
    """Batches that are not multiples of the four bootstraps of a workgroup, all three table modes and a multi-valued table, a
    trivial ciphertext (every step skipped: the other three bootstraps of its workgroup still meet their barriers) and maximal
    residues: every output word equal to the oracle's."""
    prm = toy(beta_bsk=beta)
    ctx, o = nat.Context(prm, seed=4), orc.Oracle(prm, seed=4)
    tv = ctx.tvset(TABLES)
    for B in (1, 2, 3, 4, 5, 7, 21, 64, 301, 600):
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
        want = "k_blind_rotate_pairs_k2<10,%d>" % (1 if B <= 256 else 2 if B <= 512 else 4)
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


def test_unsupported_k2_shapes_are_refused(nat):
    from tfhe_fbs_map_amd import FbsError
    for kw in (dict(log_n_poly=11), dict(bsk_group=1), dict(l_bsk=2, beta_bsk=10), dict(k=3)):
        with pytest.raises(FbsError):
            nat.Context(toy(**kw), seed=1)


def test_eval_takes_k2_for_wide_levels_only():
    """`LutExecEnv.eval`: a program whose levels average a round of bootstraps or more runs on the k = 2 set (and decrypts to the
    reference's goldens); the same program on a few samples stays on the k = 1 set with its one-bootstrap-per-CU kernels."""
    from tfhe_fbs_map_amd import ExecConfig, parse_fbs
    rec = load_fixture("mul16__search_p15")
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    for T, want_k in ((200, 2), (8, 1)):
        cfg = ExecConfig(seed=9)
        ins, expect = subsample(rec, T)
        got = env.eval(ins, config=cfg)
        assert cfg.last_choice["params"].k == want_k, (T, cfg.last_choice["params"])
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
