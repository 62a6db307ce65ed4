"""Any GLWE dimension k >= 2 the reference's optimizer may hand back (`experiments/concrete.patch:163`: k, N, n, br_l, br_b) besides the one
shape with kernels of its own (k = 2, N = 1024, one level, two key bits per step: tests/test_gpu_k2.py): k_blind_rotate_glwe, k + 1 waves
per bootstrap -- word for word against the oracle, through programs, and at a 128-bit set the selector picks."""
import numpy as np
import pytest

from oracle import lut_oracle, tfhe_oracle as orc
from tests.helpers import load_fixture, oracle_eval_program, subsample

pytestmark = pytest.mark.gpu

TABLES = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1], [0, 1, 2, 3, 2, 1, 0], [1, 1, 1, 0, 1, 0, 0, 1, 1, 1]]


@pytest.fixture(scope="module")
def nat():
    from tfhe_fbs_map_amd import _native
    return _native


def toy(**kw):
    from tfhe_fbs_map_amd import Params
    base = dict(n=12, log_n_poly=9, k=3, l_bsk=2, beta_bsk=9, t_ksk=8, gamma_ksk=2, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=1)
    base.update(kw)
    return Params(**base)


SHAPES = [  # (log_n, k, l, beta, key bits per step)
    (9, 3, 1, 20, 1), (9, 3, 2, 9, 2), (9, 2, 3, 7, 1), (9, 2, 1, 21, 2), (9, 4, 2, 10, 1), (9, 4, 1, 18, 2),
    (8, 2, 2, 8, 1), (8, 3, 4, 5, 2), (8, 4, 1, 16, 1), (8, 4, 3, 6, 2),
    (10, 2, 2, 10, 2), (10, 2, 1, 21, 1), (10, 2, 3, 7, 1), (10, 3, 1, 20, 2), (10, 3, 2, 9, 1),
]


@pytest.mark.parametrize("log_n, k, l, beta, group", SHAPES)
def test_ragged_batches_bit_exact(nat, log_n, k, l, beta, group):
    """Batches that are not multiples of a workgroup's bootstraps, all three table modes and a multi-valued table, a trivial ciphertext
    (every step skipped: the other bootstraps of its workgroup still meet their barriers) and maximal residues: every output word equal
    to the oracle's, at one to four gadget levels, with one and with two key bits per step."""
    prm = toy(log_n_poly=log_n, k=k, l_bsk=l, beta_bsk=beta, bsk_group=group)
    ctx, o = nat.Context(prm, seed=4), orc.Oracle(prm, seed=4)
    tv = ctx.tvset(TABLES)
    want = "k_blind_rotate_glwe<%d,%d,%d,1>" % (log_n, k + 1, group)       # (up to one bootstrap per CU: one per workgroup)
    for B in (1, 2, 3, 5, 13, 70):
        msgs = np.arange(B) % 7
        ids = (np.arange(B) % 4).astype(np.uint32)
        msgs[ids == 1] = np.arange(B)[ids == 1] % 14
        msgs[ids == 3] = np.arange(B)[ids == 3] % 10
        cts = ctx.encrypt(msgs, 3 + B)
        if B > 2:
            cts[B - 1, :-1] = 0
            cts[B // 2, :] = orc.Q - 1
            # every other step skipped (both rotation amounts of the pair zero): the landing sets take turns by the steps EXECUTED
            n = prm.n
            skip = np.arange(n).reshape(-1, 2)[1::2].reshape(-1) if group == 2 else np.arange(1, n, 2)
            cts[0, skip] = 0
            cts[1, skip[: len(skip) // 2]] = 0
        ctx.profile(True)
        ctx.profile_read(reset=True)
        got = ctx.bootstrap_batch(tv, cts, ids)
        assert want in ctx.profile_kernels(), (want, ctx.profile_kernels())
        ref, _ = o.bootstrap_batch(cts, TABLES, ids)
        assert got.shape == ref.shape == (B, k * (1 << log_n) + 1)
        assert np.array_equal(got, ref), B
    ctx.close()


@pytest.mark.parametrize("k, group", [(3, 2), (2, 1)])
def test_launch_shapes_by_size_and_the_cut(nat, k, group):
    """N = 512: one bootstrap per workgroup up to one per CU, two up to two, the throughput shape (three at k = 3, four at k = 2) beyond;
    a launch longer than a round of the throughput shape whose rest is small is CUT -- whole rounds, then the rest in the small shape.
    Each size against the oracle on both ends, either side of the cut and a spread; then the same sizes with one shape forced."""
    prm = toy(k=k, bsk_group=group, l_bsk=1, beta_bsk=18)
    ctx, o = nat.Context(prm, seed=9), orc.Oracle(prm, seed=9)
    cus, full = ctx.stat("cu_count"), 12 // (k + 1)
    name = lambda fpw: "k_blind_rotate_glwe<9,%d,%d,%d>" % (k + 1, group, fpw)
    rng = np.random.default_rng(3)
    tv = ctx.tvset(TABLES)
    cases = [(cus - 3, [name(1)]), (2 * cus - 5, [name(2)]), (full * cus - 7, [name(full)]), (full * cus + 100, [name(full), name(1)]),
             (full * cus + 2 * cus - 1, [name(full), name(2)]), (2 * full * cus + 2 * cus + 9, [name(full)])]
    for B, want in cases:
        ids = rng.integers(0, 4, B).astype(np.uint32)
        msgs = np.array([rng.integers(0, len(TABLES[i])) for i in ids])
        cts = ctx.encrypt(msgs, nonce0=11)
        cts[B - 1, :-1] = 0
        ctx.profile(True)
        ctx.profile_read(reset=True)
        got = ctx.bootstrap_batch(tv, cts, ids)
        assert sorted(k_ for k_ in ctx.profile_kernels() if "blind_rotate" in k_) == sorted(want), (B, ctx.profile_kernels())
        edge = full * cus * (B // (full * cus))
        pick = np.unique(np.clip(np.concatenate([np.arange(6), np.arange(B - 6, B), np.arange(edge - 4, edge + 4), rng.integers(0, B, 8)]), 0, B - 1))
        ref, _ = o.bootstrap_batch(cts[pick], TABLES, ids[pick])
        assert np.array_equal(got[pick], ref), B
    B = 2 * cus + 9
    ids = rng.integers(0, 4, B).astype(np.uint32)
    cts = ctx.encrypt(np.array([rng.integers(0, len(TABLES[i])) for i in ids]), nonce0=77)
    auto = ctx.bootstrap_batch(tv, cts, ids)
    for fpw in (1, 2, full):
        ctx.tune(br_glwe_fpw=fpw)
        ctx.profile_read(reset=True)
        assert np.array_equal(ctx.bootstrap_batch(tv, cts, ids), auto)          # the same ciphertexts whatever the shape
        assert name(fpw) in ctx.profile_kernels()
    ctx.close()


@pytest.mark.parametrize("p, norm2", [(4, 2), (7, 10)])
def test_default_k3_sets_at_full_size_against_the_oracle(nat, p, norm2):
    """What `LutExecEnv.eval` runs by default for p <= 8 at ordinary norms: the selector's own k = 3, N = 512 set at full n through
    fbs_bootstrap_batch_dev -- two full rounds of the throughput shape (1 536 bootstraps: three per workgroup), word for word against the
    oracle on ciphertexts that sit in every sub-slot of the first, a middle and the last workgroup, trivial ciphertexts beside
    ordinary ones; 1 024, which the launcher CUTS into a round and a launch of one bootstrap per workgroup, checked at both ends and
    either side of the cut; 500 (two per workgroup) and 200 (one)."""
    import torch
    from tfhe_fbs_map_amd.params import DEFAULT_GLWE_DIMS, choose_params, margin_sigmas, security_bits
    prm = choose_params(p, norm2, glwe_dims=DEFAULT_GLWE_DIMS)
    assert prm.k == 3 and prm.N == 512 and prm.bsk_group == 2 and prm.l_bsk == 1, "the selector moved: pin this test's parameter set"
    assert security_bits(prm) >= 127.9 and margin_sigmas(prm, norm2) >= 6.0
    ctx, o = nat.Context(prm, seed=1), orc.Oracle(prm, seed=1)
    cus = ctx.stat("cu_count")
    rng = np.random.default_rng(7 + p)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    tv = ctx.tvset(tables)
    name = lambda fpw: "k_blind_rotate_glwe<9,4,2,%d>" % fpw
    for B, pick, want in ((6 * cus, [0, 1, 2, 3, 4, 5, 3 * cus - 3, 3 * cus - 2, 3 * cus - 1, 3 * cus, 3 * cus + 1, 6 * cus - 3, 6 * cus - 2, 6 * cus - 1, 301, 1000], [name(3)]),
                          (4 * cus, [0, 2, 3 * cus - 1, 3 * cus, 3 * cus + 1, 4 * cus - 2, 4 * cus - 1, 500, 900], [name(1), name(3)]),
                          (500, [0, 1, 2, 3, 250, 498, 499], [name(2)]), (200, [0, 1, 100, 199], [name(1)])):
        msgs = rng.integers(0, p, B)
        ids = (np.arange(B) % 16).astype(np.uint32)
        cts = ctx.encrypt(msgs, nonce0=100)
        trivial = [pick[1], pick[-2]]
        cts[trivial, :-1] = 0
        d_in = torch.from_numpy(cts.view(np.int64)).cuda()
        d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
        d_out = torch.empty_like(d_in)
        ctx.profile(True)
        ctx.profile_read(reset=True)
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
        ctx.sync()
        assert sorted(k for k in ctx.profile_kernels() if "blind_rotate" in k) == sorted(want), (B, ctx.profile_kernels())
        got = d_out.cpu().numpy().view(np.uint64)
        ref, _ = o.bootstrap_batch(cts[pick], tables, ids[pick])
        assert np.array_equal(got[pick], ref), B
        keep = np.ones(B, bool)
        keep[trivial] = False
        assert np.array_equal(ctx.decrypt(got)[keep], np.array([tables[i][m] for i, m in zip(ids, msgs)])[keep])
    ctx.close()


def test_noise_model_holds_at_k3(nat):
    """The variance model the selector rests on, at the k = 3 set for (7, 10): the measured bootstrap OUTPUT noise (the blind
    rotation's term: k + 1 = 4 key polynomials' noise per step of two key bits, rounding seen through a key of k N = 1536 bits)
    against params.variances, every output far inside its box; and -- the modulus switch onto 2N = 1024 slots is the term that
    sets n at this size -- a linear combination of squared norm 10 of bootstrap outputs, bootstrapped again, decrypts."""
    from tfhe_fbs_map_amd.params import DEFAULT_GLWE_DIMS, choose_params, variances
    prm = choose_params(7, 10, glwe_dims=DEFAULT_GLWE_DIMS)
    assert prm.k == 3 and prm.N == 512
    ctx, o = nat.Context(prm, seed=13), orc.Oracle(prm, seed=13)
    rng = np.random.default_rng(5)
    table = [0] + [int(v) for v in rng.integers(0, 2, 6)]
    B = 600
    msgs = rng.integers(0, 7, B)
    cts = ctx.encrypt(msgs, nonce0=900)
    out = ctx.bootstrap_batch(ctx.tvset([table]), cts)
    assert np.array_equal(ctx.decrypt(out), [table[m] for m in msgs])
    phase = o.phase(out).astype(object)
    want = np.array([table[m] for m in msgs], dtype=object) * (2 * o.delta_half)
    err = np.array([min((int(p) - int(w)) % orc.Q, (int(w) - int(p)) % orc.Q) for p, w in zip(phase, want)], dtype=np.float64)
    predicted = np.sqrt(variances(prm)[0]) * orc.Q
    measured = float(np.sqrt(np.mean(err ** 2)))
    assert 0.5 * predicted < measured < 1.25 * predicted, (measured, predicted)
    assert err.max() < 0.1 * orc.Q / (4 * 7)
    # 3 x b0 + b1 (squared norm 10) of bootstrap outputs, bootstrapped again
    bits = np.array([table[m] for m in msgs])
    idx = rng.integers(0, B, (2, B))
    coefs = np.array([3, 1])
    value = (coefs[:, None] * bits[idx]).sum(0)                      # 0 .. 4 < 7
    lc = np.zeros_like(out)
    for c, row in zip(coefs, idx):
        lc = (lc + int(c) * out[row].astype(object)) % orc.Q
    again = ctx.bootstrap_batch(ctx.tvset([list(range(7))]), lc.astype(np.uint64))
    assert np.array_equal(ctx.decrypt(again), value)
    ctx.close()


def run_program(nat, prm, name, T, fuse=False):
    from tfhe_fbs_map_amd import parse_fbs
    rec = load_fixture(name)
    ops, outs = lut_oracle.read_fbs(rec["fbs"])
    ctx, o = nat.Context(prm, seed=6), orc.Oracle(prm, seed=6)
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    ins, expect = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=9)
    assert cts.shape[-1] == prm.k * prm.N + 1
    prog = nat.Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                       low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"], fuse_tables=fuse)
    got = prog.eval(cts, T)
    wires = oracle_eval_program(o, ops, outs, {n: cts[i] for i, n in enumerate(low["input_names"])}, fuse=fuse)
    for j, (out_name, src) in enumerate(outs):
        if src in ("0", "1"):
            assert ctx.decrypt(got[j]).tolist() == [int(src)] * T
        else:
            assert np.array_equal(got[j], wires[src]), out_name
            assert np.array_equal(ctx.decrypt(got[j]), expect[out_name])
    ctx.close()
    return prog


@pytest.mark.parametrize("name, T", [("adder8__search_p7", 3), ("demo_fbs_exec_env", 2), ("edge_outputs", 5)])
@pytest.mark.parametrize("k, log_n, group", [(3, 9, 2), (2, 10, 1), (4, 8, 1)])
def test_program_ciphertexts_bit_exact(nat, name, T, k, log_n, group):
    """Whole programs on such a context (ciphertexts of k N + 1 words through k_lincomb, the key switch over k N coefficients, the
    level calls, wire slots) against the oracle evaluating the same instruction list one ciphertext at a time: every output word
    identical, every output decrypting to the reference's cleartext result."""
    rec = load_fixture(name)
    ops, _ = lut_oracle.read_fbs(rec["fbs"])
    p = max(7, max(len(op[3]) for op in ops if op[0] == "boot"))
    run_program(nat, toy(log_n_poly=log_n, k=k, l_bsk=2, beta_bsk=9, bsk_group=group, p_msg=p), name, T)


@pytest.mark.parametrize("k, log_n, group", [(3, 9, 2), (2, 10, 1)])
def test_shared_rotations(nat, k, log_n, group):
    """Several tables on one blind rotation (FBS_LOAD_FUSE_TABLES) on such a context: the raw accumulator of (k + 1) N words out of
    k_blind_rotate_glwe, k_multi_extract over k mask polynomials -- word for word the oracle's fused evaluation."""
    prog = run_program(nat, toy(log_n_poly=log_n, k=k, l_bsk=2, beta_bsk=9, bsk_group=group, p_msg=7), "adder8__basic_p2", 3, fuse=True)
    assert prog.n_rotations < prog.n_bootstrap


@pytest.mark.parametrize("fuse", [False, True])
def test_levels_longer_than_a_round_are_cut_the_same_ciphertexts(nat, fuse):
    """A program whose levels are longer than a round of the throughput shape (3 x CUs bootstraps at k = 3) with a small rest: the
    launcher cuts every such level -- whole rounds, then the rest as a launch of its own, with shared rotations into the rest's own
    accumulator rows -- and every output ciphertext equals what two evaluations of half the samples each (no level reaches a round)
    give; the cut really happened (both launch shapes in the profile)."""
    from tfhe_fbs_map_amd import parse_fbs
    rec = load_fixture("adder8__basic_p2")
    prm = toy(l_bsk=1, beta_bsk=18, bsk_group=2, p_msg=7)
    ctx = nat.Context(prm, seed=6)
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    prog = nat.Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                       low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"], fuse_tables=fuse)
    cus = ctx.stat("cu_count")
    widest = max(prog.level_width)
    T = (3 * cus + 2 * cus - 8) // widest                      # the widest level: a round and a rest of up to two bootstraps per CU
    assert widest * T > 3 * cus and widest * T - 3 * cus <= 2 * cus and widest * (T // 2) <= 3 * cus
    ins, expect = subsample(rec, 8)
    reps = -(-T // 8)
    cts = ctx.encrypt(np.stack([np.tile(ins[n], reps)[:T] for n in low["input_names"]]), nonce0=9)
    ctx.profile(True)
    ctx.profile_read(reset=True)
    whole = prog.eval(cts, T)
    kernels = [k for k in ctx.profile_kernels() if "blind_rotate" in k]
    assert "k_blind_rotate_glwe<9,4,2,3>" in kernels and any(k.endswith(",1>") or k.endswith(",2>") for k in kernels), kernels
    h = T // 2
    halves = np.concatenate([prog.eval(np.ascontiguousarray(cts[:, :h]), h), prog.eval(np.ascontiguousarray(cts[:, h:]), T - h)], axis=1)
    assert np.array_equal(whole, halves)
    for j, name in enumerate(low["out_names"]):
        if low["out_wire"][j] >= 0:
            assert np.array_equal(ctx.decrypt(whole[j])[:8], expect[name]), name
    ctx.close()


@pytest.mark.parametrize("k, log_n, l, group", [(3, 9, 1, 2), (2, 9, 2, 1)])
def test_imported_keys_at_other_glwe_dimensions(nat, k, log_n, l, group):
    """fbs_import_keys on such a set: the oracle's keys under another seed (rows (component, level), k mask columns and the body; three
    samples per pair of key bits with two key bits per step: include/fbs_exec.h) are accepted and bootstrap exactly as the oracle does; a
    key with the body and a mask column exchanged is refused by the decryption check."""
    from tfhe_fbs_map_amd import FbsError
    prm = toy(log_n_poly=log_n, k=k, l_bsk=l, beta_bsk=18 // l, bsk_group=group)
    o, ctx = orc.Oracle(prm, seed=78), nat.Context(prm, seed=5, keygen=False)
    keys = o.keys()
    swapped = keys["bsk"].reshape(-1, k + 1, prm.N)[:, ::-1].copy().reshape(-1)
    with pytest.raises(FbsError, match="does not decrypt"):
        ctx.import_keys(keys["sk_lwe"], keys["sk_glwe"], swapped, keys["ksk"])
    ctx.import_keys(**keys)
    cts = o.encrypt(np.arange(7), nonce0=4)
    assert np.array_equal(ctx.bootstrap_batch(ctx.tvset([TABLES[0]]), cts), o.bootstrap_batch(cts, [TABLES[0]], None)[0])
    ctx.close()


def test_unbuilt_shapes_are_refused_with_a_code(nat):
    """k = 4 at N = 1024, k = 5, k >= 2 at N = 2048: no kernel -- an error code and a message at context creation, not a launch."""
    from tfhe_fbs_map_amd import FbsError, Params
    for kw in (dict(k=4, log_n_poly=10), dict(k=5, log_n_poly=8), dict(k=2, log_n_poly=11), dict(k=3, log_n_poly=12)):
        with pytest.raises(FbsError):
            nat.Context(Params(n=8, l_bsk=2, beta_bsk=8, t_ksk=4, gamma_ksk=4, p_msg=7, sigma_lwe=1 << 6, sigma_glwe=1 << 4, **kw), seed=1)
