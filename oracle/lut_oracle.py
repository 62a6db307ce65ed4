"""Cleartext restatement of the reference's FBS-program semantics.  TEST INFRASTRUCTURE ONLY.

Follows `LutExecEnv.eval` (reference fbs_mapper/fbs_exec_env.py:208-229): wires "0"/"1" are the
constants (:209), an input is its bit vector (:213-214), a LinearProd is sum(coef * wire) + const
(:215-217), a Bootstrap is a per-sample table lookup (:218-220).  Has its own tiny reader of the `.fbs`
text (format of `print`, :158-168) so that it shares no code with the product's parser.
Pinned: tests/test_oracle_cleartext.py checks it against every fixture in tests/golden/.
"""
import re

import numpy as np

_BOOT = re.compile(r"^(\S+) = Bootstrap\((\S+), \[(.*)\]\)$")
_OUT = re.compile(r"^Output (\S+) = (\S+)$")
_ENTRY = re.compile(r"^(?:np\.int64\()?(-?\d+)\)?$")   # numpy>=2 leaks np.int64(..) reprs into tables


def read_fbs(text):
    """-> (ops, outputs); ops are ('lin', name, [(coef, src)], const) / ('boot', name, src, table)."""
    ops, outputs = [], []
    for line in text.splitlines():
        line = line.rstrip()
        if not line:
            continue
        m = _OUT.match(line)
        if m:
            outputs.append((m.group(1), m.group(2)))
            continue
        m = _BOOT.match(line)
        if m:
            table = [int(_ENTRY.match(tok.strip()).group(1)) for tok in m.group(3).split(",")]
            ops.append(("boot", m.group(1), m.group(2), table))
            continue
        name, rhs = line.split(" = ", 1)
        if rhs.startswith("Input("):
            continue
        terms, const = [], 0
        for piece in rhs.split(" + "):
            piece = piece.strip()
            if " * " in piece:
                c, src = piece.split(" * ")
                terms.append((int(c), src))
            elif piece:
                const += int(piece)
        ops.append(("lin", name, terms, const))
    return ops, outputs


def evaluate(ops, outputs, input_values, all_wires=False):
    wires = {"0": 0, "1": 1}
    for k, v in input_values.items():
        wires[k] = np.asarray(v).reshape(-1).astype(np.int64)
    for op in ops:
        if op[0] == "lin":
            _, name, terms, const = op
            acc = const
            for c, src in terms:
                acc = acc + c * wires[src]
            wires[name] = acc
        else:
            _, name, src, table = op
            wires[name] = np.asarray(table, dtype=np.int64)[wires[src]]
    if all_wires:
        return wires
    return {name: wires[src] for name, src in outputs}


def eval_fbs_text(text, input_values):
    ops, outputs = read_fbs(text)
    return evaluate(ops, outputs, input_values)
