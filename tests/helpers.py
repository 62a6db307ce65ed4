"""Shared by the test modules: fixture loading and an oracle-backed program evaluator."""
import glob
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture_names(pattern="*"):
    names = (os.path.basename(p)[:-len(".json.gz")] for p in glob.glob(os.path.join(GOLDEN, pattern + ".json.gz")))
    return sorted(n for n in names if not n.startswith("_"))        # _*.json.gz: collections, not one program each


def _decode(v):
    if "const" in v:
        return int(v["const"])
    if "bits" in v:
        raw = np.frombuffer(bytes.fromhex(v["bits"]), np.uint8)
        return np.unpackbits(raw)[:v["n"]].astype(np.int64)
    return np.asarray(v["ints"], np.int64)


_cache = {}


def load_fixture(name):
    if name not in _cache:
        with gzip.open(os.path.join(GOLDEN, name + ".json.gz"), "rb") as f:
            rec = json.loads(f.read().decode())
        rec["inputs"] = {k: _decode(v) for k, v in rec["inputs"].items()}
        rec["outputs"] = {k: _decode(v) for k, v in rec["outputs"].items()}
        if rec.get("outputs_bitenv"):
            rec["outputs_bitenv"] = {k: _decode(v) for k, v in rec["outputs_bitenv"].items()}
        _cache[name] = rec
    return _cache[name]


def toy_k2(p_msg=7, n=12, beta=21):
    """A toy set of the k = 2 shape (GLWE dimension 2 at N = 1024, one gadget level, two key bits per step): n small enough for
    the CPU oracle to bootstrap in milliseconds, real N so that the kernels are the shipped ones (k_blind_rotate_pairs_k2 and
    the whole-CU k = 2 shape, ct_words = 2049 through k_lincomb and the level calls)."""
    from tfhe_fbs_map_amd import Params
    return Params(n=n, log_n_poly=10, k=2, l_bsk=1, beta_bsk=beta, t_ksk=8, gamma_ksk=2, p_msg=p_msg, sigma_lwe=1 << 8, sigma_glwe=4,
                  bsk_group=2)


def toy_k3(p_msg=7, n=12, beta=18):
    """A toy set of the k = 3 shape the default 128-bit sets for p <= 8 run on (GLWE dimension 3 at N = 512, one gadget level, two key
    bits per step: k_blind_rotate_glwe, ciphertexts of 3 N + 1 = 1537 words, accumulator rows of 4 N)."""
    from tfhe_fbs_map_amd import Params
    return Params(n=n, log_n_poly=9, k=3, l_bsk=1, beta_bsk=beta, t_ksk=8, gamma_ksk=2, p_msg=p_msg, sigma_lwe=1 << 8, sigma_glwe=4,
                  bsk_group=2)


def toy_glwe(k, p_msg=7, n=12):
    """toy_k2 / toy_k3 by GLWE dimension"""
    return toy_k2(p_msg, n) if k == 2 else toy_k3(p_msg, n)


def subsample(rec, T):
    """First T samples of the harness inputs and of the expected outputs."""
    ins = {k: v[:T] for k, v in rec["inputs"].items()}
    outs = {k: (v if isinstance(v, int) else v[:T]) for k, v in rec["outputs"].items()}
    return ins, outs


def assert_outputs_equal(got, expected):
    assert set(got.keys()) == set(expected.keys())
    for k, e in expected.items():
        g = got[k]
        if isinstance(e, int):
            assert isinstance(g, (int, np.integer)) or np.ndim(g) == 0, k
            assert int(g) == e, k
        else:
            assert np.array_equal(np.asarray(g).reshape(-1), e), "output %s differs" % k


def oracle_eval_program(orc, ops, outputs, in_cts, fuse=False):
    """Run a program (oracle.lut_oracle.read_fbs form) on ciphertexts with the C oracle.
    in_cts: {input name: [T][ct_words]}.  Returns {wire name: [T][ct_words]} for all wires.
    fuse: a wire that two or more bootstraps read is rotated once and every table cut out of that accumulator
    (oracle bootstrap_multi) -- what a program loaded with FBS_LOAD_FUSE_TABLES computes."""
    wires = dict(in_cts)
    T = len(next(iter(in_cts.values()))) if in_cts else 1
    readers = {}
    for op in ops:
        if op[0] == "boot":
            readers.setdefault(op[2], []).append(op)
    for op in ops:
        if op[0] == "lin":
            _, name, terms, const = op
            out = np.empty((T, orc.ctw), np.uint64)
            for s in range(T):
                out[s] = orc.lincomb([wires[src][s] for _, src in terms], [c for c, _ in terms], const)
            wires[name] = out
        elif fuse and len(readers[op[2]]) >= 2:
            if op[1] in wires:
                continue                                        # computed with the first gate of its source
            group = readers[op[2]]
            res = orc.bootstrap_multi(wires[op[2]], [g[3] for g in group])
            for g, r in zip(group, res):
                wires[g[1]] = r
        else:
            _, name, src, table = op
            res, _ = orc.bootstrap_batch(wires[src], [table], None)
            wires[name] = res
    return wires
