"""Boolean netlists on the caller side of the executor: a gate-level circuit container, BLIF and Bristol readers,
and the one-gate-one-bootstrap lowering to a `LutExecEnv` program.

This is the "front door" SURVEY.md 8(f)1 asks for, so that netlists the reference consumes (EPFL / ISCAS BLIF,
Bristol-fashion MPC circuits) reach the GPU executor without the reference installed:

* `BitExecEnv` mirrors the reference's container of the same name (bit_exec_env.py:5-194: `input`, `output`,
  `op_lut`, `op_not/and/xor/or` with constant folding, `eval`, `print`, `stats`, `remove_dangling_nodes`,
  `to_blif` :247-279) -- same method names and printed forms, own implementation (flat tuples, numpy evaluation).
* `parse_blif` reads the BLIF subset the reference accepts through PyPI `blifparser` (map_circuit.py:12-50):
  `.model/.inputs/.outputs/.names/.end`, `\\` continuations, `#` comments; single-output covers whose rows all carry
  the same output value; rows index the table most-significant input first (map_circuit.py:12-23).  Beyond the
  reference: `-` don't-cares in cubes and more than two inputs per cover are accepted.
* `parse_bristol` reads Bristol-fashion text (the reference goes through PyPI `bfcl`, map_circuit.py:53-88):
  gates AND / XOR / OR / INV / NOT / EQW / EQ.
* `map_basic` is the reference's `MapToFBSBasic` lowering (map_to_fbs.py:15-52): NOT becomes the levelled
  `-x + 1`, a buffer is a wire, any other k-input gate becomes `sum 2^(k-1-i) x_i` followed by one bootstrap with
  the gate's truth table.  Its printed program is compared with the reference's, fixture by fixture, in
  tests/test_netlist.py.
"""
import io
import sys

import numpy as np

from .fbs_exec_env import LutExecEnv

__all__ = ["BitExecEnv", "parse_blif", "parse_bristol", "map_basic"]


class BitExecEnv:
    """Gate-level circuit: inputs, k-input look-up tables over earlier nodes, named outputs."""

    class Node:
        __slots__ = ("name",)

        def __init__(self, name):
            self.name = name

    class Const(Node):
        __slots__ = ("val",)

        def __init__(self, val):
            super().__init__(str(val))
            self.val = val

        def __str__(self):
            return self.name

    class Input(Node):
        def __str__(self):
            return "Input(%s)" % self.name

    class LUT(Node):
        __slots__ = ("inputs", "truth_table", "kind")
        _FORMS = {"and": "AND(%s, %s)", "xor": "XOR(%s, %s)", "or": "OR(%s, %s)", "not": "Not(%s)"}

        def __init__(self, name, inputs, truth_table, kind=None):
            super().__init__(name)
            self.inputs = list(inputs)
            self.truth_table = truth_table
            self.kind = kind

        def __str__(self):
            names = tuple(i.name for i in self.inputs)
            if self.kind:
                return self._FORMS[self.kind] % names
            return "LUT([%s], %s)" % (", ".join(names), self.truth_table)

    CONST0 = Const(0)
    CONST1 = Const(1)

    def __init__(self):
        self.instructions = []
        self.inputs = []
        self.outputs = {}
        self.ids = set()
        self._counter = 0

    # ---- construction ---------------------------------------------------------------------------------------
    def _name_for(self, name):
        if name is None:
            while True:
                self._counter += 1
                name = "n%d" % self._counter
                if name not in self.ids:
                    break
        else:
            assert name not in self.ids, "id already exists in circuit"
        self.ids.add(name)
        return name

    def input(self, input_id):
        node = BitExecEnv.Input(input_id)
        self.instructions.append(node)
        self.inputs.append(node)
        return node

    def output(self, name, node):
        assert isinstance(node, BitExecEnv.Node), "Expected BitExecEnv.Node"
        self.outputs[name] = node

    def _gate(self, inputs, table, name, kind=None):
        node = BitExecEnv.LUT(self._name_for(name), inputs, table, kind)
        self.instructions.append(node)
        return node

    def op_lut(self, inputs, truth_table, name=None):
        assert 2 ** len(inputs) == len(truth_table), "length miss-match"
        assert all(isinstance(i, BitExecEnv.Node) for i in inputs), "Error"
        assert min(truth_table) == 0 and max(truth_table) == 1, "truth table wrong values"
        return self._gate(inputs, truth_table, name)

    def op_not(self, inp, name=None):
        if inp is BitExecEnv.CONST0:
            return BitExecEnv.CONST1
        if inp is BitExecEnv.CONST1:
            return BitExecEnv.CONST0
        return self._gate([inp], [1, 0], name, "not")

    def _binary(self, kind, table, x, y, name, fold):
        # `fold(const_value, other)` is what the gate reduces to when one side is a constant
        for c, other in ((x, y), (y, x)):
            if isinstance(c, BitExecEnv.Const):
                return fold(c.val, other)
        assert x.name != y.name, "something is wrong"
        return self._gate([x, y], table, name, kind)

    def op_and(self, inp1, inp2, name=None):
        return self._binary("and", [0, 0, 0, 1], inp1, inp2, name, lambda c, o: o if c else BitExecEnv.CONST0)

    def op_xor(self, inp1, inp2, name=None):
        return self._binary("xor", [0, 1, 1, 0], inp1, inp2, name, lambda c, o: self.op_not(o) if c else o)

    def op_or(self, inp1, inp2, name=None):
        return self._binary("or", [0, 1, 1, 1], inp1, inp2, name, lambda c, o: BitExecEnv.CONST1 if c else o)

    # ---- inspection -----------------------------------------------------------------------------------------
    def print(self, os=sys.stdout, show_inputs=True, show_outputs=True):
        for node in self.instructions:
            if isinstance(node, BitExecEnv.Input) and not show_inputs:
                continue
            print("%s = %s" % (node.name, node), file=os)
        if show_outputs:
            for name, node in self.outputs.items():
                print("Output %s = %s" % (name, node.name), file=os)

    def eval(self, input_values):
        """Cleartext evaluation on T samples per input (bit_exec_env.py:173-194): a gate looks its table up at the
        index formed by its inputs, first input most significant."""
        wires = {"0": 0, "1": 1}
        for node in self.instructions:
            if isinstance(node, BitExecEnv.Input):
                wires[node.name] = np.array(input_values[node.name]).reshape(-1)
            else:
                index = 0
                for src in node.inputs:
                    index = index * 2 + wires[src.name]
                wires[node.name] = np.asarray(node.truth_table, dtype=int)[index]
        return {name: wires[node.name] for name, node in self.outputs.items()}

    def remove_dangling_nodes(self):
        live = {node.name for node in self.outputs.values()}
        for node in reversed(self.instructions):
            if node.name in live and isinstance(node, BitExecEnv.LUT):
                live.update(src.name for src in node.inputs)
        self.instructions = [node for node in self.instructions if node.name in live]

    def stats(self):
        count = dict(nb_inp=0, nb_and=0, nb_xor=0, nb_not=0, nb_lut=0, max_lut_inputs=0, max_lut_size=0)
        for node in self.instructions:
            if isinstance(node, BitExecEnv.Input):
                count["nb_inp"] += 1
            elif node.kind in ("and", "xor", "not"):
                count["nb_" + node.kind] += 1
            else:   # plain tables and OR gates (the reference has no OR counter, bit_exec_env.py:206-233)
                count["nb_lut"] += 1
                count["max_lut_inputs"] = max(count["max_lut_inputs"], len(node.inputs))
                count["max_lut_size"] = max(count["max_lut_size"], len(node.truth_table))
        count["nb_out"] = len(self.outputs)
        return count

    def to_blif(self, fs=sys.stdout, model_name="test"):
        """BLIF text in the reference's layout (bit_exec_env.py:247-279): each cover lists the minority rows."""
        out = [".model %s" % model_name,
               ".inputs %s" % " ".join(node.name for node in self.inputs),
               ".outputs %s" % " ".join(self.outputs.keys())]
        for node in self.instructions:
            if isinstance(node, BitExecEnv.Input):
                continue
            table = node.truth_table
            bits = len(node.inputs)
            listed = 1 if 2 * sum(table) <= len(table) else 0
            out.append(".names %s %s" % (" ".join(src.name for src in node.inputs), node.name))
            rows = ["%s %d" % (format(i, "0%db" % bits), listed) for i, v in enumerate(table) if v == listed]
            out.append("\n".join(rows))
        for name, node in self.outputs.items():
            if node.name != name:
                out.append(".names %s %s\n1 1" % (node.name, name))
        out.append(".end")
        print("\n".join(out), file=fs)


# ---------------------------------------------------------------------------------------------------------------
# readers
# ---------------------------------------------------------------------------------------------------------------
def _text_of(source):
    if isinstance(source, io.IOBase) or hasattr(source, "read"):
        return source.read()
    if "\n" not in source and not source.lstrip().startswith("."):
        with open(source) as f:
            return f.read()
    return source


def _blif_statements(text):
    """Logical lines: comments stripped, `\\` continuations joined, blank lines dropped."""
    pending = ""
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].rstrip()
        if line.endswith("\\"):
            pending += line[:-1] + " "
            continue
        line = (pending + line).strip()
        pending = ""
        if line:
            yield line
    if pending.strip():
        yield pending.strip()


def _cover_table(rows, n_inputs, where):
    """Truth table of a single-output cover, first input most significant (map_circuit.py:12-23)."""
    if not rows:
        return [0] * (1 << n_inputs)          # an empty cover is the constant 0
    phases = {r[-1] for r in rows}
    assert len(phases) == 1 and phases <= {"0", "1"}, "%s: rows of one cover must share one output value" % where
    listed = int(phases.pop())
    table = [1 - listed] * (1 << n_inputs)
    for r in rows:
        cube = r[0] if n_inputs else ""
        assert len(cube) == n_inputs, "%s: cube width does not match the input list" % where
        free = [i for i, ch in enumerate(cube) if ch == "-"]
        fixed = sum(1 << (n_inputs - 1 - i) for i, ch in enumerate(cube) if ch == "1")
        assert all(ch in "01-" for ch in cube), "%s: bad cube %r" % (where, cube)
        for fill in range(1 << len(free)):
            k = fixed
            for j, i in enumerate(free):
                if (fill >> j) & 1:
                    k |= 1 << (n_inputs - 1 - i)
            table[k] = listed
    return table


def parse_blif(source):
    """BLIF text, path or file object -> BitExecEnv.  Constants fold into CONST0/CONST1 (map_circuit.py:38-41)."""
    env = BitExecEnv()
    wires = {}
    outputs = []
    covers = []            # (signal names, rows) in file order
    current = None
    for line in _blif_statements(_text_of(source)):
        if line.startswith("."):
            words = line.split()
            key = words[0]
            current = None
            if key == ".inputs":
                for name in words[1:]:
                    wires[name] = env.input(name)
            elif key == ".outputs":
                outputs += words[1:]
            elif key == ".names":
                current = (words[1:], [])
                covers.append(current)
            elif key in (".model", ".end"):
                pass
            else:
                raise AssertionError("unsupported BLIF construct %r" % key)
        else:
            assert current is not None, "cover row outside a .names block: %r" % line
            current[1].append(line.split())
    # BLIF does not promise definition-before-use; resolve covers in dependency order
    by_output = {}
    for sig, rows in covers:
        assert sig[-1] not in wires and sig[-1] not in by_output, "signal %r is driven twice" % sig[-1]
        by_output[sig[-1]] = (sig, rows)
    state = {}

    def build(name):
        if name in wires:
            return wires[name]
        assert name in by_output, "signal %r is never defined" % name
        assert state.get(name) != "open", "combinational loop through %r" % name
        state[name] = "open"
        stack = [(name, iter(by_output[name][0][:-1]))]
        while stack:
            top, deps = stack[-1]
            pushed = False
            for d in deps:
                if d in wires:
                    continue
                assert d in by_output, "signal %r is never defined" % d
                assert state.get(d) != "open", "combinational loop through %r" % d
                state[d] = "open"
                stack.append((d, iter(by_output[d][0][:-1])))
                pushed = True
                break
            if pushed:
                continue
            stack.pop()
            sig, rows = by_output[top]
            table = _cover_table(rows, len(sig) - 1, ".names " + " ".join(sig))
            if len(set(table)) == 1:
                wires[top] = BitExecEnv.CONST1 if table[0] else BitExecEnv.CONST0
            else:
                wires[top] = env.op_lut([wires[s] for s in sig[:-1]], table, name=top)
            state[top] = "done"
        return wires[name]

    for sig, _ in covers:
        build(sig[-1])
    for name in outputs:
        env.output(name, build(name))
    return env


_BRISTOL_TABLES = {"AND": [0, 0, 0, 1], "XOR": [0, 1, 1, 0], "OR": [0, 1, 1, 1], "NAND": [1, 1, 1, 0],
                   "NOR": [1, 0, 0, 0], "XNOR": [1, 0, 0, 1], "EQ2": [1, 0, 0, 1]}


def parse_bristol(source):
    """Bristol-fashion circuit text -> BitExecEnv.  Inputs are named `i_<wire>`, outputs by wire index, as in the
    reference (map_circuit.py:53-88).  Header: `gates wires`, `niv n_1 .. n_niv`, `nov m_1 .. m_nov`; the output
    wires are the last sum(m) wires.  The older two-line header (`n1 n2 n3`) is accepted too."""
    lines = [ln.split() for ln in _text_of(source).splitlines() if ln.strip()]
    n_gates, n_wires = int(lines[0][0]), int(lines[0][1])
    second = [int(x) for x in lines[1]]
    if len(lines[2]) >= 2 and lines[2][-1].isalpha():       # old format: gates start on the third line
        n_in = second[0] + second[1]
        n_out = second[2]
        first_gate = 2
    else:
        assert second[0] == len(second) - 1, "bad input-value line"
        third = [int(x) for x in lines[2]]
        assert third[0] == len(third) - 1, "bad output-value line"
        n_in, n_out = sum(second[1:]), sum(third[1:])
        first_gate = 3
    env = BitExecEnv()
    wires = {i: env.input("i_%d" % i) for i in range(n_in)}
    gates = lines[first_gate:first_gate + n_gates]
    assert len(gates) == n_gates, "gate count does not match the header"
    for g in gates:
        fan_in, fan_out = int(g[0]), int(g[1])
        assert fan_out == 1, "multi-output gates are not supported"
        ins = [int(x) for x in g[2:2 + fan_in]]
        out = int(g[2 + fan_in])
        op = g[-1].upper()
        if op in ("INV", "NOT"):
            wires[out] = env.op_not(wires[ins[0]], name="w_%d" % out)
        elif op in ("EQW", "BUF"):
            wires[out] = wires[ins[0]]
        elif op == "EQ":                                     # constant assignment: `1 1 <0|1> out EQ`
            wires[out] = BitExecEnv.CONST1 if ins[0] else BitExecEnv.CONST0
        else:
            assert op in _BRISTOL_TABLES and fan_in == 2, "unsupported gate %r" % g
            a, b = wires[ins[0]], wires[ins[1]]
            fold = {"AND": env.op_and, "XOR": env.op_xor, "OR": env.op_or}.get(op)
            if fold is not None and (isinstance(a, BitExecEnv.Const) or isinstance(b, BitExecEnv.Const)):
                wires[out] = fold(a, b)
            elif isinstance(a, BitExecEnv.Const) or isinstance(b, BitExecEnv.Const) or a is b:
                raise AssertionError("constant or repeated operand on %s gate not supported" % op)
            else:
                wires[out] = env.op_lut([a, b], _BRISTOL_TABLES[op], name="w_%d" % out)
    for w in range(n_wires - n_out, n_wires):
        env.output(w, wires[w])
    return env


# ---------------------------------------------------------------------------------------------------------------
# lowering
# ---------------------------------------------------------------------------------------------------------------
def map_basic(env: BitExecEnv) -> LutExecEnv:
    """One bootstrap per gate (map_to_fbs.py:15-52).  A k-input gate needs message space 2^k, so programs mapped
    this way run at p >= 4 for two-input netlists (the executor takes tables up to 2p when they are negacyclic)."""
    lut_env = LutExecEnv()
    wires = {"0": lut_env.const(0), "1": lut_env.const(1)}
    for node in env.instructions:
        if isinstance(node, BitExecEnv.Input):
            wires[node.name] = lut_env.input(node.name)
            continue
        table = list(node.truth_table)
        srcs = [wires[s.name] for s in node.inputs]
        assert len(table) == 1 << len(srcs)
        if len(srcs) == 1:
            if table == [1, 0]:
                wires[node.name] = lut_env.linear([-1], srcs, const_coef=1)
            else:
                assert table == [0, 1]
                wires[node.name] = srcs[0]
        else:
            weights = [1 << (len(srcs) - 1 - i) for i in range(len(srcs))]
            wires[node.name] = lut_env.bootstrap(lut_env.linear(weights, srcs), table)
    for name, node in env.outputs.items():
        lut_env.output(name, wires[node.name])
    return lut_env
