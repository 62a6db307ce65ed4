"""`python -m tfhe_fbs_map_amd FILE` -- evaluate a circuit on ciphertexts on the GPU the way the reference's harness
evaluates it in the clear (fbs_mapper/map_circuit.py:124-188).

FILE is a mapped program the reference wrote (`.fbs` from `--output`, `.lbf` from `--output_lbf`) or a gate netlist
(`.blif`, Bristol fashion), which is lowered one gate per bootstrap (`netlist.map_basic`, the reference's
`--mapper basic`).  Inputs are the harness's: `np.random.seed(42)`, one `randint(0, 2, T)` draw per input in program
order (map_circuit.py:137-139).  For netlists the decrypted outputs are compared with the netlist's own cleartext
evaluation -- the reference's self-check (:174-180) with ciphertexts in the middle.  One JSON line is printed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np


def load(path, kind, inputs):
    from . import parse_fbs, parse_lbf
    from .netlist import map_basic, parse_blif, parse_bristol
    text = open(path).read()
    if kind == "auto":
        ext = os.path.splitext(path)[1].lower()
        kind = {".fbs": "fbs", ".lbf": "lbf", ".blif": "blif", ".txt": "bristol", ".bristol": "bristol"}.get(ext)
        if kind is None:
            kind = "blif" if ".model" in text else "lbf" if ".lincomb" in text or ".bootstrap" in text else "fbs"
    if kind == "fbs":
        return parse_fbs(text, inputs=inputs), None, kind
    if kind == "lbf":
        return parse_lbf(text), None, kind
    bits = parse_blif(text) if kind == "blif" else parse_bristol(text)
    bits.remove_dangling_nodes()
    return map_basic(bits), bits, kind


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m tfhe_fbs_map_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("filename")
    ap.add_argument("--type", choices=["auto", "fbs", "lbf", "blif", "bristol"], default="auto")
    ap.add_argument("--samples", type=int, default=1000, help="samples per input (the harness uses 1000)")
    ap.add_argument("--fbs_size", type=int, default=None, help="plaintext modulus p (default: smallest that fits)")
    ap.add_argument("--inputs", default=None,
                    help="comma-separated input names in harness order (.fbs files do not list their inputs)")
    ap.add_argument("--seed", type=int, default=None, help="key-generation seed (default: fresh from os.urandom)")
    ap.add_argument("--reduced-noise", action="store_true",
                    help="benchmark parameter set with reduced noise (NOT secure) instead of the 128-bit selector")
    ap.add_argument("--no-shared-rotations", action="store_true",
                    help="give every table a blind rotation of its own, even where several read one linear combination")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    from . import ExecConfig
    names = args.inputs.split(",") if args.inputs else None
    env, bits, kind = load(args.filename, args.type, names)
    order = [i.name for i in bits.inputs] if bits is not None else [
        i.name for i in env.instructions if isinstance(i, type(env).Input)]
    np.random.seed(42)
    values = {name: np.random.randint(0, 2, (args.samples)) for name in order}

    cfg = ExecConfig(fbs_size=args.fbs_size, seed=args.seed, device=args.device, reduced_noise=args.reduced_noise,
                     fuse_tables=False if args.no_shared_rotations else None)
    stats = env.stats()
    t0 = time.perf_counter()
    out = env.eval(values, config=cfg)          # first call: key generation + upload + program load + run
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = env.eval(values, config=cfg)
    steady = time.perf_counter() - t0
    result = dict(file=args.filename, type=kind, samples=args.samples, stats=stats,
                  first_eval_s=round(first, 3), eval_s=round(steady, 3),
                  fbs_per_s=round(stats["nb_bootstrap"] * args.samples / steady, 1) if steady > 0 else None,
                  outputs={str(k): (int(v) if np.ndim(v) == 0 else int(np.asarray(v).sum())) for k, v in out.items()})
    from dataclasses import asdict
    from .params import margin_sigmas, security_bits
    for prog, _ in cfg._programs.values():
        ctx = prog.ctx
        norm2 = env.fusion_stats(ctx.params.p_msg)["norm2_linprod"] if prog.fused else stats["norm2_linprod"]
        result["params"] = asdict(ctx.params)
        result["security_bits_estimate"] = round(security_bits(ctx.params), 1)
        result["margin_sigmas"] = round(margin_sigmas(ctx.params, norm2), 2)
        result["blind_rotations"] = prog.n_rotations          # < nb_bootstrap where tables share a rotation
        result["shared_rotations"] = prog.fused
    if bits is not None:
        clear = bits.eval(values)
        result["matches_cleartext_netlist"] = all(
            np.array_equal(np.broadcast_to(np.asarray(out[k]), (args.samples,)),
                           np.broadcast_to(np.asarray(clear[k]), (args.samples,))) for k in clear)
    print(json.dumps(result))
    return 0 if result.get("matches_cleartext_netlist", True) else 1


if __name__ == "__main__":
    sys.exit(main())
