"""Every kernel instantiation the launchers can pick, one by one, against the CPU oracle.

`fbs_kernel_catalog` lists them by the rules of the dispatch (csrc/fbs_blind_rotate.hip, fbs_blind_rotate_cu.hip,
fbs_kernels.hip).  For each name this module knows a parameter set, a batch size and the launcher knobs that lead to it; it
runs the launch, asserts that the launcher really took that instantiation (the per-kernel profile table), and compares the
output ciphertexts with the oracle's word for word -- all of them for small batches, a subsample that covers both ends and
the crafted ciphertexts for large ones.  A catalog entry without a recipe here fails the suite, so a new launcher branch
cannot ship unchecked."""
import re

import numpy as np
import pytest

from oracle import tfhe_oracle as orc

pytestmark = pytest.mark.gpu

TABLES = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 2, 3, 2, 1, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1]]
CUS = 256

# what the launcher derives DIG from (dev_blind_rotate): l > 5 -> 0; l = 1 -> 4; l = 2 -> 4 + by_beta; else by_beta,
# by_beta = 3 for beta <= 7, 2 for beta <= 9, 1 otherwise
GADGET_OF_DIG = {0: (6, 4), 1: (3, 10), 2: (3, 8), 3: (3, 7), 4: (1, 20), 5: (2, 10), 6: (2, 8), 7: (2, 7)}


def recipe(name):
    """-> dict(log_n, l, beta, group, count, knobs) that makes the launcher pick `name`"""
    m = re.fullmatch(r"k_blind_rotate<(\d+),(\d+),(\d+),(\d+)(,false)?>", name)
    if m:
        L, LL, dig, fpw = (int(m.group(i)) for i in range(1, 5))
        l, beta = GADGET_OF_DIG[dig]
        main_ll = 6 if L <= 10 else L - 4
        knobs = {}
        if m.group(5):                                    # no priority hand-over: two-level N = 1024 sets beyond two rounds
            count = 8 * CUS + 60
        elif fpw == 4:                                    # whole rounds of the benchmark shape
            count = 4 * CUS
        elif fpw == 2:                                    # between one and two bootstraps per CU, two per workgroup
            count, knobs = CUS + 44, dict(br_cu_kernel=0)
        elif LL != main_ll:                               # N = 1024 / 2048 on four waves per polynomial, generic kernel
            count, knobs = 40, dict(br_cu_kernel=0)
        elif L in (10, 11):                               # main shape: more than two bootstraps per CU (below, the one-per-CU shapes)
            count = 2 * CUS + 88
        else:
            count = 40
        return dict(log_n=L, l=l, beta=beta, group=1, count=count, knobs=knobs)
    m = re.fullmatch(r"k_blind_rotate_pairs<(\d+),(\d+),(\d+)>", name)
    if m:
        L, dig = int(m.group(1)), int(m.group(3))
        l, beta = {4: (1, 20), 3: (2, 7), 0: (2, 10)}[dig]
        # (N = 2048 with two levels goes to the whole-CU kernel whatever the size: the A/B switch brings the generic one back)
        return dict(log_n=L, l=l, beta=beta, group=2, count=CUS + 40 if (L, l) == (11, 1) else 40,
                    knobs=dict(br_cu_kernel=0) if (L, l) == (11, 2) else {})
    if name == "k_blind_rotate_cu_pairs<11,1>":           # two key bits per step on a whole CU: up to one bootstrap per CU
        return dict(log_n=11, l=1, beta=20, group=2, count=CUS - 9, knobs={})
    if name == "k_blind_rotate_cu_pairs<11,2>":           # ... with two gadget levels: every launch, round after round
        return dict(log_n=11, l=2, beta=10, group=2, count=CUS + 21, knobs={})
    m = re.fullmatch(r"k_blind_rotate_pairs_k2<10,(\d)>", name)
    if m:                                                 # GLWE dimension k = 2 on three waves per bootstrap: one, two or four bootstraps per
        fpw = int(m.group(1))                             # workgroup, a ragged last one (up to three per CU the launcher prefers the shape below)
        return dict(log_n=10, l=1, beta=20, group=2, k=2, count={1: 41, 2: CUS + 41, 4: 3 * CUS + 41}[fpw],
                    knobs={} if fpw == 4 else dict(br_k2_shape=3))
    if name == "k_blind_rotate_cu_k2":                    # ... one bootstrap on the twelve waves of a workgroup: two rounds, the second partial
        return dict(log_n=10, l=1, beta=20, group=2, k=2, count=CUS + 41, knobs={})
    m = re.fullmatch(r"k_blind_rotate_glwe<(\d+),(\d),(\d),(\d)>", name)
    if m:                                                 # every other GLWE dimension / size / depth: k + 1 waves per bootstrap, two gadget levels
        L, k1, group, fpw = (int(m.group(i)) for i in (1, 2, 3, 4))   # (k = 2 at N = 1024 with one level and two key bits per step has its own kernels);
        # one bootstrap per workgroup up to one per CU, two up to two, the throughput shape beyond: a ragged last workgroup each time
        return dict(log_n=L, l=2, beta=8, group=group, k=k1 - 1, count={1: 41, 2: CUS + 41}.get(fpw, 2 * CUS + 41), knobs={})
    m = re.fullmatch(r"k_blind_rotate_cu<(\d+),(\d+),(\d+)(,lean)?>", name)
    if m:
        L, nl, first = int(m.group(1)), int(m.group(2)), int(m.group(3))
        beta = {2: 7, 1: 9, 0: 10}[first]
        if nl * beta > 30:
            beta = 30 // nl
        if m.group(4):                                    # two workgroups per CU: between one and two bootstraps per CU
            return dict(log_n=L, l=nl, beta=beta, group=1, count=CUS + 70, knobs=dict(br_cu_lean=1))
        return dict(log_n=L, l=nl, beta=beta, group=1, count=CUS + 3 if nl == 3 or (L, nl) == (11, 2) else 40, knobs=dict(br_cu_lean=0))
    ks = {"k_ks_gemm<2,2> (int8 MFMA)": (40, {}), "k_keyswitch_fp<8,2,8>": (70, dict(ks_mfma=0)),
          "k_keyswitch_lanes<8,2,8>": (70, dict(ks_mfma=0, ks_fp=0)), "k_keyswitch_lanes<8,1,4>": (40, dict(ks_mfma=0)),
          "k_keyswitch<8>": (9, dict(ks_mfma=0))}
    if name in ks:
        count, knobs = ks[name]
        return dict(log_n=9, l=2, beta=8, group=1, count=count, knobs=knobs)
    return None


def catalog():
    from tfhe_fbs_map_amd import _native
    return _native.kernel_catalog()


def pytest_generate_tests(metafunc):
    if "kernel_name" in metafunc.fixturenames:
        metafunc.parametrize("kernel_name", catalog())


def run_case(name, rec):
    from tfhe_fbs_map_amd import Params, _native as nat
    n = 8 if rec["log_n"] < 12 else 4
    prm = Params(n=n, log_n_poly=rec["log_n"], k=rec.get("k", 1), l_bsk=rec["l"], beta_bsk=rec["beta"], t_ksk=4, gamma_ksk=4 if rec["log_n"] < 12 else 3, p_msg=7,
                 sigma_lwe=1 << 6, sigma_glwe=1 << 4, bsk_group=rec["group"])
    ctx, o = nat.Context(prm, seed=21), orc.Oracle(prm, seed=21)
    ctx.tune(**rec["knobs"])
    count = rec["count"]
    rng = np.random.default_rng(count)
    ids = rng.integers(0, len(TABLES), count).astype(np.uint32)
    msgs = np.array([rng.integers(0, len(TABLES[i])) for i in ids])
    cts = ctx.encrypt(msgs, nonce0=3)
    crafted = sorted({1 % count, count // 2, count - 2})
    cts[crafted[0], :-1] = 0                                 # a trivial ciphertext: every rotation amount is zero
    cts[crafted[-1], :] = orc.Q - 1                          # maximal residues
    ctx.profile(True)
    ctx.profile_read(reset=True)
    got = ctx.bootstrap_batch(ctx.tvset(TABLES), cts, ids)
    launched = ctx.profile_kernels()
    assert name in launched, (name, sorted(launched))
    if count <= 64:
        pick = np.arange(count)
    else:                                                    # both ends, the crafted ones, every sub-slot of the first workgroups, a spread
        pick = np.unique(np.concatenate([np.arange(8), np.arange(count - 8, count), crafted, rng.integers(0, count, 12)]))
    ref, _ = o.bootstrap_batch(cts[pick], TABLES, ids[pick])
    assert np.array_equal(got[pick], ref), name
    assert got.max() < orc.Q
    ctx.close()


def test_every_catalog_entry_has_a_recipe():
    missing = [k for k in catalog() if recipe(k) is None]
    assert not missing, missing
    assert len(set(catalog())) == len(catalog()) >= 160


def test_instantiation_against_the_oracle(kernel_name):
    run_case(kernel_name, recipe(kernel_name))


def test_cu_count_matches_the_recipes():
    from tfhe_fbs_map_amd import Params, _native as nat
    ctx = nat.Context(Params(n=4, log_n_poly=8, p_msg=7), seed=1)
    assert ctx.stat("cu_count") == CUS, "the batch sizes of the recipes assume 256 CUs"
