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


def test_unbuilt_shapes_are_refused_with_a_code(nat):
    """k = 4 at N = 1024, k = 5, k >= 2 at N = 2048: no kernel -- an error code and a message at context creation, not a launch."""
    from tfhe_fbs_map_amd import FbsError, Params
    for kw in (dict(k=4, log_n_poly=10), dict(k=5, log_n_poly=8), dict(k=2, log_n_poly=11), dict(k=3, log_n_poly=12)):
        with pytest.raises(FbsError):
            nat.Context(Params(n=8, l_bsk=2, beta_bsk=8, t_ksk=4, gamma_ksk=4, p_msg=7, sigma_lwe=1 << 6, sigma_glwe=1 << 4, **kw), seed=1)
