"""Known-answer vectors that FREEZE the ciphertext conventions of this repository.

The reference has no cryptography, so the rounding, decomposition, key-derivation and noise conventions of
DESIGN.md section 2 are this repository's own, and the oracle (oracle/tfhe_oracle.c) and the kernels are edited by the same
hands.  `decrypt == the reference's goldens` catches wrong results, not a silent change of conventions made on both sides
at once.  This script writes SHA-256 digests of what the ORACLE produces for fixed seeds and parameter sets:

    python tests/golden/make_ciphertext_kats.py            # rewrites tests/golden/_ciphertext_kats.json

tests/test_oracle_tfhe.py::test_ciphertext_kats (CPU) holds the oracle to them, tests/test_gpu_parity.py::
test_ciphertext_kats_on_the_gpu the kernels.  A convention change must therefore re-run this script, commit the new file
in the SAME commit, and say why in the commit message and in DESIGN.md section 3.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import lut_oracle, tfhe_oracle as orc                     # noqa: E402
from tests.helpers import load_fixture, oracle_eval_program, subsample  # noqa: E402

OUT = os.path.join(HERE, "_ciphertext_kats.json")

# parameter sets are written out (P1024, and what params.choose_params returned for (15, 70), (31, 325), (4, 2) at the end of
# round 2), not asked of the selector: its choice may move, the conventions must not
SETS = {
    "p1024": dict(n=630, log_n_poly=10, k=1, l_bsk=3, beta_bsk=7, t_ksk=8, gamma_ksk=2, p_msg=15, sigma_lwe=64, sigma_glwe=64,
                  bsk_group=1),
    "secure_p15_two_key_bits_per_step": dict(n=714, log_n_poly=11, k=1, l_bsk=1, beta_bsk=21, t_ksk=7, gamma_ksk=2, p_msg=15,
                                             sigma_lwe=1065975446, sigma_glwe=4, bsk_group=2),
    "secure_p31": dict(n=766, log_n_poly=11, k=1, l_bsk=2, beta_bsk=14, t_ksk=8, gamma_ksk=2, p_msg=31, sigma_lwe=408668278,
                       sigma_glwe=4, bsk_group=1),
    # (round 3: what the selector returns for (31, 325) since two key bits per step with two gadget levels have their kernel)
    "secure_p31_two_key_bits_per_step": dict(n=766, log_n_poly=11, k=1, l_bsk=2, beta_bsk=14, t_ksk=8, gamma_ksk=2, p_msg=31,
                                             sigma_lwe=408668278, sigma_glwe=4, bsk_group=2),
    # (round 3) GLWE dimension k = 2 at N = 1024, two key bits per step (k_blind_rotate_pairs_k2): n = 760 and a 16-bit key switch
    # give p = 15 at norm2 70 6.7 sigma by params.variances.  Frozen from the oracle (general in k) before the kernel existed.
    "secure_p15_k2_n1024_two_key_bits_per_step": dict(n=760, log_n_poly=10, k=2, l_bsk=1, beta_bsk=21, t_ksk=8, gamma_ksk=2, p_msg=15,
                                                      sigma_lwe=456472211, sigma_glwe=4, bsk_group=2),
    # (round 4) the k = 2 set the selector SHIPS for (15, 70) -- n = 734, a 14-bit key switch -- in a batch long enough (3 x 256 CUs
    # + 5) for the launcher to take the throughput shape k_blind_rotate_pairs_k2<10,4> (four bootstraps per workgroup; up to three
    # bootstraps per CU it prefers the twelve-waves-per-bootstrap shape); LAYOUTS below says which of its 773 ciphertexts are ordinary.  The n = 760 entry above stays: it is the freeze that predates the kernel.
    "secure_p15_k2_shipped_n734_four_per_workgroup": dict(n=734, log_n_poly=10, k=2, l_bsk=1, beta_bsk=21, t_ksk=7, gamma_ksk=2, p_msg=15,
                                                          sigma_lwe=737229119, sigma_glwe=4, bsk_group=2),
    # (end of round 4) GLWE dimension k = 3 at N = 512, two key bits per step (k_blind_rotate_glwe): the set the selector ships for
    # (4, 2) once glwe_dims = (1, 2, 3) is the default -- in a batch of 2 x 256 CUs + 6, long enough for the launcher to take the
    # throughput shape (three bootstraps per workgroup); LAYOUTS says which ciphertexts are ordinary
    "secure_p4_k3_n512_shipped_three_per_workgroup": dict(n=614, log_n_poly=9, k=3, l_bsk=1, beta_bsk=18, t_ksk=11, gamma_ksk=1, p_msg=4,
                                                          sigma_lwe=6737066039, sigma_glwe=280, bsk_group=2),
    "secure_p4_n1024": dict(n=638, log_n_poly=10, k=1, l_bsk=2, beta_bsk=8, t_ksk=12, gamma_ksk=1, p_msg=4, sigma_lwe=4328098537,
                            sigma_glwe=3511592, bsk_group=1),
}
# sets frozen AHEAD of their kernels would go here (held by the CPU test only); none at present
SETS_AHEAD = {}
SEED = 1
COUNT = 5
# name -> (batch size, positions of the ORDINARY ciphertexts): every other ciphertext of the batch is trivial (mask zero, so every
# blind-rotation step is skipped: milliseconds in the oracle, and on the GPU a bootstrap that only keeps its workgroup's barriers
# company).  The ordinary ones sit in all four sub-slots of the first workgroup, in a middle one, and in the ragged last one.
LAYOUTS = {"secure_p15_k2_shipped_n734_four_per_workgroup": (773, [0, 1, 2, 3, 386, 769, 771, 772]),
           # (three bootstraps per workgroup: every sub-slot of the first one, a middle one, the ragged last one of 518 = 172 x 3 + 2)
           "secure_p4_k3_n512_shipped_three_per_workgroup": (518, [0, 1, 2, 259, 515, 516, 517])}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def batch_case(name, prm):
    p = prm["p_msg"]
    rng = np.random.default_rng(42)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(4)]
    tables.append([int(v) for v in rng.integers(0, p, p)])                 # multi-valued
    count, ordinary = LAYOUTS.get(name, (COUNT, list(range(COUNT - 1))))
    trivial = [i for i in range(count) if i not in set(ordinary)]
    msgs = [int(v) for v in rng.integers(0, p, count)]
    ids = [i % len(tables) for i in range(count)]
    o = orc.Oracle(prm, seed=SEED)
    cts = o.encrypt(np.array(msgs), nonce0=7)
    cts[trivial, :-1] = 0                                                  # trivial ciphertexts: the r == 0 path of every step
    out, _ = o.bootstrap_batch(cts, tables, np.array(ids, np.uint32))
    keys = o.keys()
    small = np.stack([o.modswitch(o.keyswitch(ct)) for ct in cts[:2]])
    return dict(kind="batch", params=prm, seed=SEED, nonce0=7, tables=tables, msgs=msgs, table_ids=ids, trivial=trivial,
                sha256=dict(sk_lwe=digest(keys["sk_lwe"]), sk_glwe=digest(keys["sk_glwe"]), bsk_first_row=digest(keys["bsk"][:2 << prm["log_n_poly"]]),
                            ksk_first_row=digest(keys["ksk"][:prm["n"] + 1]), inputs=digest(cts), modswitched_first_two=digest(small),
                            outputs=digest(out)),
                decrypts_to=[int(v) for v in o.decrypt(out)])


def fused_case():
    rec = load_fixture("adder8__basic_p2")
    ops, outs = lut_oracle.read_fbs(rec["fbs"])
    prm = dict(n=12, log_n_poly=10, k=1, l_bsk=3, beta_bsk=7, t_ksk=8, gamma_ksk=2, p_msg=7, sigma_lwe=256, sigma_glwe=256, bsk_group=1)
    o = orc.Oracle(prm, seed=6)
    T = 2
    ins, _ = subsample(rec, T)
    names = list(rec["program_inputs"])
    cts = o.encrypt(np.stack([ins[n] for n in names]), nonce0=9)
    wires = oracle_eval_program(o, ops, outs, {n: cts[i] for i, n in enumerate(names)}, fuse=True)
    res = np.stack([wires[src] for _, src in outs if src not in ("0", "1")])
    return dict(kind="fused_program", fixture="adder8__basic_p2", params=prm, seed=6, nonce0=9, samples=T,
                sha256=dict(inputs=digest(cts), outputs=digest(res)))


def main():
    kats = {name: batch_case(name, prm) for name, prm in SETS.items()}
    kats["fused_adder8"] = fused_case()
    ahead = {name: batch_case(name, prm) for name, prm in SETS_AHEAD.items()}
    with open(OUT, "w") as f:
        json.dump(dict(modulus=int(orc.Q), note="written by tests/golden/make_ciphertext_kats.py from oracle/tfhe_oracle.c; "
                                                "see its docstring before changing this file", kats=kats,
                       kats_ahead_of_the_kernels=ahead), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
