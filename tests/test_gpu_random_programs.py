"""Randomly generated programs through the drop-in builder, evaluated on ciphertexts on the GPU, against the
cleartext oracle on the text the builder prints.  Covers shapes the mappers never emit: multi-valued tables,
negative coefficients, constants in combinations, lincomb outputs, tables that use the negacyclic half (lengths
between p and 2p in all three of the reference's modes, map_to_fbs.py:81-98), deep chains and wide fan-in."""
import io

import numpy as np
import pytest

from oracle import lut_oracle
from tests.helpers import assert_outputs_equal

pytestmark = pytest.mark.gpu
P = 15


def random_table(rng, length):
    """A table of `length` entries evaluable at p = P with minimum 0: any values when it fits the half torus,
    otherwise values whose sum with the entry p places further on is one constant (table[i] + table[i+p] == c)."""
    if length <= P:
        t = [int(v) for v in rng.integers(0, int(rng.integers(1, 4)) + 1, length)]
        t[int(rng.integers(0, length))] = 0
        return t
    c = int(rng.integers(0, 3))                  # the reference's modes 2, 1 and 3: c = 0, 1, 2
    t = [0] * length
    for i in range(1, P):
        t[i] = int(rng.integers(0, c + 1))
    for i in range(length - P):
        t[i + P] = c - t[i]
    return t


def random_program(seed):
    from tfhe_fbs_map_amd import LutExecEnv, table_is_valid
    rng = np.random.default_rng(seed)
    env = LutExecEnv()
    nodes = [env.input("i%d" % k) for k in range(int(rng.integers(3, 7)))]
    boots = 0
    for _ in range(int(rng.integers(12, 40))):
        fan = int(rng.integers(1, 5))
        picks = [nodes[int(rng.integers(0, len(nodes)))] for _ in range(fan)]
        coefs = [int(rng.integers(-2, 4)) or 1 for _ in range(fan)]
        const = int(rng.integers(0, 3))
        # shift so that the minimum is 0 (what the mappers do, map_to_fbs.py:270-275)
        lo = sum(min(0, c * env.max_val[v.name]) for c, v in zip(coefs, picks))
        lin = env.linear(coefs, picks, const_coef=const - lo)
        width = env.max_val[lin.name] + 1
        if width > 2 * P:
            continue
        table = random_table(rng, width)
        if min(table) != 0 or not table_is_valid(table, P):
            continue
        nodes.append(env.bootstrap(lin, table))
        boots += 1
        if rng.random() < 0.2:
            nodes.append(lin)                    # a levelled value used again without a bootstrap
    for k, node in enumerate(nodes[-4:]):
        env.output("o%d" % k, node)
    env.output("k", env.const(int(rng.integers(0, 2))))
    return env, boots


@pytest.mark.parametrize("seed,k", [(s, 1) for s in range(12)] + [(s, 2) for s in (0, 3, 5, 8)] + [(s, 3) for s in (1, 4, 7, 9)])
def test_random_program(seed, k):
    """k = 2, 3: the same programs on a context of GLWE dimension 2 at N = 1024 / 3 at N = 512 (ciphertexts of k N + 1 words through the
    linear combinations, the level calls and k_blind_rotate_pairs_k2 / k_blind_rotate_glwe)."""
    from tfhe_fbs_map_amd import ExecConfig
    from tests.helpers import toy_glwe
    env, boots = random_program(seed)
    if boots == 0:
        pytest.skip("generator produced no bootstrap")
    rng = np.random.default_rng(1000 + seed)
    names = [i.name for i in env.instructions if isinstance(i, type(env).Input)]
    ins = {n: rng.integers(0, 2, 24) for n in names}
    buf = io.StringIO()
    env.print(os=buf, show_outputs=True)
    expect = lut_oracle.eval_fbs_text(buf.getvalue(), ins)
    cfg = ExecConfig(fbs_size=P, seed=1, reduced_noise=True) if k == 1 else ExecConfig(fbs_size=P, seed=1, params=toy_glwe(k, P, n=40))
    got = env.eval(ins, config=cfg)
    assert cfg.last_choice["params"].k == k
    assert_outputs_equal(got, {name: (int(v) if np.ndim(v) == 0 else np.asarray(v, np.int64)) for name, v in expect.items()})


def random_shared_program(seed):
    """As above, but most linear combinations get two to four tables (distinct ones: the builder merges identical tables on
    one source, fbs_exec_env.py:93-100) -- the shape that shares blind rotations."""
    from tfhe_fbs_map_amd import LutExecEnv, table_is_valid
    rng = np.random.default_rng(5000 + seed)
    env = LutExecEnv()
    nodes = [env.input("i%d" % k) for k in range(int(rng.integers(3, 6)))]
    shared = 0
    for _ in range(int(rng.integers(8, 20))):
        fan = int(rng.integers(1, 4))
        picks = [nodes[int(rng.integers(0, len(nodes)))] for _ in range(fan)]
        coefs = [int(rng.integers(-2, 4)) or 1 for _ in range(fan)]
        lo = sum(min(0, c * env.max_val[v.name]) for c, v in zip(coefs, picks))
        lin = env.linear(coefs, picks, const_coef=int(rng.integers(0, 2)) - lo)
        width = env.max_val[lin.name] + 1
        if width > 2 * P or width < 2:
            continue
        made = 0
        for _ in range(int(rng.integers(1, 5))):
            table = random_table(rng, width)
            if min(table) != 0 or max(table) == 0 or not table_is_valid(table, P):
                continue
            nodes.append(env.bootstrap(lin, table))
            made += 1
        shared += made >= 2
    for k, node in enumerate(nodes[-6:]):
        env.output("o%d" % k, node)
    return env, shared


@pytest.mark.parametrize("seed", range(8))
def test_random_program_with_shared_sources(seed):
    """Several random tables (multi-valued, all negacyclic modes) per linear combination, evaluated with shared blind
    rotations: every output equals the cleartext oracle's, and fewer rotations ran than there are tables."""
    from tfhe_fbs_map_amd import ExecConfig
    env, shared = random_shared_program(seed)
    if shared == 0:
        pytest.skip("generator produced no shared source")
    rng = np.random.default_rng(2000 + seed)
    names = [i.name for i in env.instructions if isinstance(i, type(env).Input)]
    ins = {n: rng.integers(0, 2, 24) for n in names}
    buf = io.StringIO()
    env.print(os=buf, show_outputs=True)
    expect = lut_oracle.eval_fbs_text(buf.getvalue(), ins)
    cfg = ExecConfig(fbs_size=P, seed=1, reduced_noise=True, fuse_tables=True)
    got = env.eval(ins, config=cfg)
    assert_outputs_equal(got, {k: (int(v) if np.ndim(v) == 0 else np.asarray(v, np.int64)) for k, v in expect.items()})
    (prog, _), = cfg._programs.values()
    assert prog.fused and prog.n_rotations < prog.n_bootstrap
