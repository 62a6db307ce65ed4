"""Drop-in for the reference's `fbs_mapper/fbs_exec_env.py`: same class, same builder and
serialiser surface, but `eval` runs the program HOMOMORPHICALLY on an MI355X.

Reference interface mirrored here (file:line in /root/reference/fbs_mapper/fbs_exec_env.py):
  LutExecEnv(merge_linear_prods=True) :63      .input :102   .const :105   .linear :131
  .bootstrap :147   .output :154   .print :158   .write_lbf :170   .eval :208
  .remove_dangling_nodes :231   .stats :245   node classes Node/Const/Input/LinearProd/Bootstrap :12-61
  attributes instructions / outputs / max_val / instr_cache :65-69

`eval(input_values)` keeps the reference's contract -- dict of 0/1 arrays in, dict of integer
arrays out (python ints for constant outputs) -- by encrypting the inputs, running every
LinearProd / Bootstrap on ciphertexts through libfbsexec.so (ctypes, C ABI in
include/fbs_exec.h) and decrypting the outputs.  There is no cleartext or CPU path in this
module; without the GPU library it raises.

Additions with no counterpart in the reference: `parse_fbs` / `parse_lbf` (the reference only
writes those formats, :158-206), `schedule()` (bootstrap levels), `lower()` (flat program for
the C ABI) and the `ExecConfig` that picks parameters and caches keyed contexts.
The north-star name `FbsExecEnv` is an alias of `LutExecEnv`.
"""
from __future__ import annotations

import logging
import math
import os
import re
import sys
import textwrap
from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np

__all__ = ["LutExecEnv", "FbsExecEnv", "ExecConfig", "parse_fbs", "parse_lbf", "min_fbs_size", "table_is_valid"]


# --------------------------------------------------------------------------------------------
# negacyclic table contract (reference fbs_mapper/map_to_fbs.py:81-98)
# --------------------------------------------------------------------------------------------
def table_is_valid(table, p):
    """Can one functional bootstrap with plaintext modulus `p` evaluate `table`?
    Yes when it fits the half torus (len <= p), or when it is at most 2p long and the value met
    at x + p is `c - table[x]` for one constant c (c = 1, 0, 2 are the reference's three modes)."""
    L = len(table)
    if L <= p:
        return True
    if L > 2 * p:
        return False
    c = table[0] + table[p]
    return all(table[i] + table[i + p] == c for i in range(L - p))


def min_fbs_size(tables, at_least=2):
    """Smallest p for which every table is evaluable."""
    p = max(2, at_least)
    longest = max((len(t) for t in tables), default=1)
    while p < longest and not all(table_is_valid(t, p) for t in tables):
        p += 1
    return p


def table_fusion_norms(table, p):
    """(|D_F|^2, mean of G_F^2) for `table` at plaintext modulus p, where TV_F = Delta/2 G_F(X) = TV_0 D_F (include/
    fbs_exec.h, FBS_LOAD_FUSE_TABLES): D_F is non-zero where the table changes value between neighbouring boxes and where
    the last half box flips to c - f(0); G_F is 2 f - c on every box."""
    assert table_is_valid(table, p)
    L = len(table)
    c = table[0] + table[p] if L > p else 0
    f = [int(table[x]) if x < L else 0 for x in range(p)]
    steps = [f[x] - f[x - 1] for x in range(1, p)] + [c - f[0] - f[p - 1]]
    return sum(d * d for d in steps), sum((2 * v - c) ** 2 for v in f) / p


def table_fusion_factor(table, p):
    """Upper bound of the factor by which sharing a blind rotation multiplies the noise variance of this table's output.
    Key-noise part of the rotation noise: independent coefficients, |D_F|^2.  Rounding part: seen through the binary key
    S = 1/2 (1 + .. + X^(N-1)) + centred part, and (1 + .. + X^(N-1)) D_F = G_F, so it grows by |D_F|^2 / 2 + mean(G_F^2) / 2
    (measured: tests/test_gpu_fusion.py).  The larger of the two holds for any mix."""
    d2, g2 = table_fusion_norms(table, p)
    return max(1.0, float(d2), d2 / 2.0 + g2 / 2.0)


# --------------------------------------------------------------------------------------------
# execution configuration (no counterpart in the reference)
# --------------------------------------------------------------------------------------------
@dataclass
class ExecConfig:
    """How `LutExecEnv.eval` reaches the GPU.  `fbs_size=None` picks the smallest p that can
    evaluate every table of the program.  `params=None` asks `params.choose_params` for the cheapest
    parameter set that is 128-bit secure (noise from `params.sigma_min`) and leaves `min_margin`
    standard deviations of room at the program's own (p, norm2_linprod) -- the role of the patched
    optimizer in the reference's flow (experiments/add_exec_estimates.py:9-16).  Pass `params`
    explicitly (e.g. `params.P1024`, the reduced-noise benchmark set) to override.

    What "128-bit" covers: the NOISE LEVELS of the chosen set (params.sigma_min).  The randomness behind them is test-grade
    (`_native.RANDOMNESS_GRADE`: ChaCha20 streams, an Irwin-Hall stand-in for the discrete Gaussian); with `seed=None` the
    generator is keyed with 256 bits from the OS and the parameter set is mixed into the derivation, with an int seed it is
    the reproducible 64-bit form tests use.  A deployment brings its own keys: `Context.import_keys`."""
    fbs_size: int | None = None
    params: object | None = None          # tfhe_fbs_map_amd.Params (p_msg is overridden by fbs_size)
    seed: int | bytes | None = None       # key seed: int = reproducible (tests); None = 32 bytes from os.urandom, once per ExecConfig
    device: int = 0
    nonce0: int | None = None             # first encryption nonce (reproducible runs); None = streams nobody has used (the context counts)
    min_margin: float = 6.0               # p_error ~ 2e-9 per bootstrap (the reference's optimizer default is 4 sigma)
    # Where nothing reaches `min_margin` at N <= 4096 the selector raises.  Set this (e.g. params.REFERENCE_MARGIN = 4.0, the
    # reference optimizer's own default, p_error 6e-5 per bootstrap) to let it step down to that floor instead; the margin a
    # program actually got and what it means for the whole program is in `last_choice` either way.
    allow_margin_floor: float | None = None
    security: int = 128
    reduced_noise: bool = False           # params=None: the reduced-noise benchmark set for p (params.params_for) -- NOT secure
    # Several tables on one linear combination (the reference's one-gate-one-bootstrap lowering, map_to_fbs.py:41-45): share
    # ONE blind rotation per source (FBS_LOAD_FUSE_TABLES).  None = when it is cheaper: the shared outputs carry |D_F|^2
    # times the noise variance, so the parameter set is chosen for the program's FUSED norm and its cost times the rotations
    # left is compared with the unfused choice.  With explicit `params` None means off (their margin is the caller's).
    fuse_tables: bool | None = None
    # GLWE dimensions the selector may use.  k = 2 (N = 1024, two key bits per step) carries what the k = 1 sets need N = 2048 for
    # at 0.81 of their cost per bootstrap in launches of a round (1 024 bootstraps) or more (k_blind_rotate_pairs_k2), and since
    # round 4 it is ahead at EVERY launch size: one bootstrap on the twelve waves of a workgroup (k_blind_rotate_cu_k2) takes
    # 2.0-2.4 ms per launch of up to one bootstrap per CU where the k = 1 sets' whole-CU kernels take 2.6-2.9.  So the choice no
    # longer depends on how wide a program's levels are (rounds 3's `wide_level`), nor on how many ranks it is cut over: every
    # rank of a sharded run derives the same set from (p, norm2) -- shared rotations (fuse_tables) included: the accumulator rows
    # and the extraction kernel are general in k since round 4.  k = 3 (N = 512, two key bits per step: k_blind_rotate_glwe, k + 1
    # waves per bootstrap) sits between the two noise floors k N = 1024 and 2048 and is what p <= 8 takes at ordinary norms: 183-208 k
    # FBS/s in full rounds where the k = 2 sets give 151-164 k, 1.5-1.9 ms per launch of up to one bootstrap per CU where they take 2.0-2.4.
    glwe_dims: tuple = (1, 2, 3)          # (= params.DEFAULT_GLWE_DIMS)
    max_programs: int = 8                 # loaded programs kept per ExecConfig (least recently used evicted)
    _contexts: dict = field(default_factory=dict, repr=False)
    _programs: "OrderedDict" = field(default_factory=OrderedDict, repr=False)
    last_choice: dict | None = field(default=None, repr=False)   # what `choose` decided for the most recent program

    def key_seed(self):
        if self.seed is None:
            self.seed = os.urandom(32)
        return self.seed

    def params_choice(self, p, norm2=1, glwe_dims=None):
        """The parameter set a program with plaintext modulus p and noise statistic norm2 is evaluated with (`glwe_dims`: the GLWE
        dimensions admitted for it; default: this configuration's)."""
        from .params import REFERENCE_MARGIN, choose_params, params_for
        if self.params is None and self.reduced_noise:
            return params_for(p)
        if self.params is None:
            floor = None if self.allow_margin_floor is None else min(self.min_margin, self.allow_margin_floor)
            try:
                return choose_params(p, norm2, min_margin=self.min_margin, security=self.security, floor_margin=floor,
                                     glwe_dims=tuple(self.glwe_dims if glwe_dims is None else glwe_dims))
            except ValueError as e:
                if floor is not None:
                    raise
                raise ValueError("%s; ExecConfig(allow_margin_floor=%.1f) accepts the reference optimizer's own %.1f sigma "
                                 "(p_error 6.3e-5 per bootstrap)" % (e, REFERENCE_MARGIN, REFERENCE_MARGIN)) from None
        return self.params.replace(p_msg=p)

    def context_of(self, prm):
        from . import _native as nat
        key = (prm, self.key_seed(), self.device)
        ctx = self._contexts.get(key)
        if ctx is None:
            ctx = self._contexts[key] = nat.Context(prm, seed=self.key_seed(), device=self.device)
        return ctx

    def context_for(self, p, norm2=1):
        return self.context_of(self.params_choice(p, norm2))

    def choose(self, env, p, samples=None, ranks=1):
        """(context, fuse) for a program: the parameter set of `params_choice` at the program's norm, and whether the
        tables of shared sources share their blind rotation (`fuse_tables`).  `samples` / `ranks`: how many samples it is about
        to be evaluated on, over how many GPUs -- recorded in `last_choice` with the launch sizes they imply; they do not move
        the parameter set (see `glwe_dims`: k = 2 is ahead at every launch size, so one GPU and every rank of a sharded run take
        the same set).  Leaves in `last_choice` the margin the program got and the failure probability that goes with it."""
        from .params import bootstrap_cost
        stats = env.stats()
        fstats = env.fusion_stats(p) if self.fuse_tables is not False else None
        if fstats is None or fstats["nb_rotation"] == stats["nb_bootstrap"] or (self.fuse_tables is None and self.params is not None):
            return self._chosen(self.params_choice(p, stats["norm2_linprod"]), False, stats["norm2_linprod"], stats["nb_bootstrap"], samples, ranks)
        fused = self.params_choice(p, fstats["norm2_linprod"])
        if self.fuse_tables is not True:
            plain = self.params_choice(p, stats["norm2_linprod"])
            if not bootstrap_cost(fused) * fstats["nb_rotation"] < bootstrap_cost(plain) * stats["nb_bootstrap"]:
                return self._chosen(plain, False, stats["norm2_linprod"], stats["nb_bootstrap"], samples, ranks)
        return self._chosen(fused, True, fstats["norm2_linprod"], stats["nb_bootstrap"], samples, ranks)

    def _chosen(self, prm, fuse, norm2, nb_bootstrap, samples=None, ranks=1):
        from .params import margin_sigmas, p_error
        margin = margin_sigmas(prm, norm2)
        per_bootstrap = p_error(margin)
        self.last_choice = dict(params=prm, fuse_tables=fuse, norm2=norm2, margin_sigmas=margin, asked_margin=self.min_margin,
                                samples=samples, ranks=ranks,
                                p_error_per_bootstrap=per_bootstrap,
                                p_error_per_sample=-math.expm1(nb_bootstrap * math.log1p(-min(per_bootstrap, 0.5))),
                                relaxed=self.params is None and not self.reduced_noise and margin < self.min_margin - 1e-9)
        return self.context_of(prm), fuse

    def program_for(self, ctx, low, fuse=False):
        """The loaded (device-resident) form of a lowered program, cached; the cache is bounded because every entry
        pins index tables in HBM (the wire buffer itself belongs to the context and is shared)."""
        from . import _native as nat
        key = (id(ctx), id(low), bool(fuse))
        hit = self._programs.get(key)
        if hit is not None and hit[1] is low:
            self._programs.move_to_end(key)
            return hit[0]
        tv = ctx.tvset(low["tables"])
        prog = nat.Program(ctx, tv, len(low["input_names"]), low["kind"], low["arg0"], low["arg1"], low["const_coef"],
                           low["term_coef"], low["term_src"], low["out_wire"], fuse_tables=fuse)
        self._programs[key] = (prog, low)
        while len(self._programs) > max(1, self.max_programs):
            _, (old, _) = self._programs.popitem(last=False)
            old.close()
        return prog

    def take_nonces(self, count):
        """Every ciphertext ever encrypted under one ExecConfig gets its own randomness stream: None = the context hands out
        streams nobody has used (fbs_encrypt_fresh); an explicit `nonce0` pins them for reproducible runs."""
        return self.nonce0


class LutExecEnv:
    # ---- node types (names, attributes, __str__ forms as in the reference :12-61) -------------
    class Node:
        def __init__(self, name):
            self.name = name

        def __eq__(self, other):
            return repr(self) == repr(other)

        def __hash__(self):
            return hash(self.name)

    class Const(Node):
        def __init__(self, value):
            LutExecEnv.Node.__init__(self, str(value))
            self.value = value

        def __str__(self):
            return str(self.value)

    class Input(Node):
        def __str__(self):
            return "Input(%s)" % self.name

    class LinearProd(Node):
        def __init__(self, name, coef_vals, const_coef=0):
            LutExecEnv.Node.__init__(self, name)
            if not all(isinstance(v, LutExecEnv.Node) for _, v in coef_vals):
                raise AssertionError("Expected 'Node' type")
            self.coef_vals = coef_vals
            self.const_coef = const_coef

        def __str__(self):
            terms = " + ".join("%s * %s" % (c, v.name) for c, v in self.coef_vals)
            tail = "+ %s" % self.const_coef if self.const_coef != 0 else ""
            return terms + " " + tail          # the reference leaves a trailing blank when const is 0

    class Bootstrap(Node):
        def __init__(self, name, val, table):
            LutExecEnv.Node.__init__(self, name)
            if not isinstance(table, list):
                raise AssertionError("Expected list")
            if not isinstance(val, LutExecEnv.Node):
                raise AssertionError("Expected LutExecEnv.Node")
            self.table = table
            self.val = val

        def __str__(self):
            return "Bootstrap(%s, %s)" % (self.val.name, self.table)

    #: shared default; assign a new ExecConfig (class- or instance-level) to change device/params
    exec_config = ExecConfig()

    # ---- construction -------------------------------------------------------------------------
    def __init__(self, merge_linear_prods=True):
        self._unique_id = 0
        self.instructions = []
        self.outputs = {}
        self._merge_linear_prods = merge_linear_prods
        self.max_val = {}
        self.instr_cache = {}
        self.logger = logging.getLogger("LutExecEnv")
        self._lowered = None

    def _new_id(self):
        self._unique_id += 1
        return "m%d" % self._unique_id

    def _bound(self, instr):
        """Upper bound of the value an instruction can take (lower bound is 0 by construction)."""
        if isinstance(instr, LutExecEnv.Input):
            return 1
        if isinstance(instr, LutExecEnv.LinearProd):
            return instr.const_coef + sum(max(0, c * self.max_val[v.name]) for c, v in instr.coef_vals)
        if isinstance(instr, LutExecEnv.Bootstrap):
            assert min(instr.table) == 0
            return max(instr.table)
        raise AssertionError("Unknown instruction")

    def _add_instr(self, instr):
        key = str(instr)
        hit = self.instr_cache.get(key)
        if hit is not None:                       # common-subexpression: same text, same node
            return hit
        assert instr.name not in self.max_val, "Error"
        self.instr_cache[key] = instr
        self.instructions.append(instr)
        self.max_val[instr.name] = self._bound(instr)
        self._lowered = None
        return instr

    def input(self, input_id):
        return self._add_instr(LutExecEnv.Input(input_id))

    def const(self, value):
        return LutExecEnv.Const(value)

    def linear(self, coefs, vals, const_coef=0):
        terms = []
        for coef, val in zip(coefs, vals):
            assert isinstance(val, LutExecEnv.Node), "Expected LutExecEnv.Node"
            if isinstance(val, LutExecEnv.LinearProd) and self._merge_linear_prods:
                terms.extend((coef * c, v) for c, v in val.coef_vals)      # inline the inner combination
                const_coef += coef * val.const_coef
            elif isinstance(val, LutExecEnv.Const):
                const_coef += coef * val.value
            else:
                terms.append((coef, val))
        return self._add_instr(LutExecEnv.LinearProd(self._new_id(), terms, const_coef))

    def bootstrap(self, val, table):
        assert isinstance(val, LutExecEnv.Node), "Expected LutExecEnv.Node"
        assert isinstance(table, list), "Expected list"
        assert len(table) == self.max_val[val.name] + 1, "%s vs %s %s" % (table, val.name, self.max_val[val.name])
        return self._add_instr(LutExecEnv.Bootstrap(self._new_id(), val, table))

    def output(self, name, val):
        assert isinstance(val, LutExecEnv.Node), "Expected LutExecEnv.Node"
        self.outputs[name] = val
        self._lowered = None

    # ---- serialisers (formats of the reference :158-206) ----------------------------------------
    def print(self, os=sys.stdout, show_inputs=False, show_outputs=False):
        for instr in self.instructions:
            if isinstance(instr, LutExecEnv.Input) and not show_inputs:
                continue
            print("%s = %s" % (instr.name, instr), file=os)
        if show_outputs:
            for name, val in self.outputs.items():
                print("Output %s = %s" % (name, val.name), file=os)

    def write_lbf(self, os=sys.stdout):
        def wrapped(line):
            return " \\\n ".join(textwrap.wrap(line))

        names = [i.name for i in self.instructions if isinstance(i, LutExecEnv.Input)]
        print(wrapped(".inputs " + " ".join(names)), file=os)
        print(wrapped(".outputs " + " ".join(str(k) for k in self.outputs)), file=os)
        for instr in self.instructions:
            if isinstance(instr, LutExecEnv.Input):
                continue
            if isinstance(instr, LutExecEnv.LinearProd):
                ordered = sorted(instr.coef_vals, key=lambda cv: cv[1].name)
                print(".lincomb %s %s" % (" ".join(v.name for _, v in ordered), instr.name), file=os)
                const = str(instr.const_coef) if instr.const_coef != 0 else ""
                print("%s %s" % (" ".join(str(c) for c, _ in ordered), const), file=os)
            elif isinstance(instr, LutExecEnv.Bootstrap):
                print(".bootstrap %s %s" % (instr.val.name, instr.name), file=os)
                print("".join(str(t) for t in instr.table), file=os)
            else:
                raise AssertionError("Unknown instruction")
        for out, val in self.outputs.items():
            print(".lincomb %s %s" % (val.name, out), file=os)
            print("1", file=os)

    # ---- analysis --------------------------------------------------------------------------------
    def remove_dangling_nodes(self):
        live = {v.name for v in self.outputs.values()}
        for instr in reversed(self.instructions):
            if instr.name not in live:
                continue
            if isinstance(instr, LutExecEnv.LinearProd):
                live.update(v.name for _, v in instr.coef_vals)
            elif isinstance(instr, LutExecEnv.Bootstrap):
                live.add(instr.val.name)
        self.instructions = [i for i in self.instructions if i.name in live]
        self._lowered = None

    def stats(self):
        count = dict(inp=0, lin=0, boot=0)
        widest = 0
        norm2 = {}
        for instr in self.instructions:
            if isinstance(instr, LutExecEnv.Input):
                count["inp"] += 1
                norm2[instr.name] = 1
            elif isinstance(instr, LutExecEnv.LinearProd):
                count["lin"] += 1
                norm2[instr.name] = sum(c * c * norm2[v.name] for c, v in instr.coef_vals)
            elif isinstance(instr, LutExecEnv.Bootstrap):
                count["boot"] += 1
                widest = max(widest, len(instr.table))
                norm2[instr.name] = 1          # a bootstrap resets the noise
            else:
                raise AssertionError("Unknown instruction")
        return dict(nb_inp=count["inp"], nb_linprod=count["lin"], nb_bootstrap=count["boot"], max_lut_size=widest,
                    norm2_linprod=max(norm2.values()), nb_out=len(self.outputs))

    def fusion_stats(self, p):
        """What sharing blind rotations does to this program at plaintext modulus p (no counterpart in the reference):
        nb_rotation = blind rotations left when every wire that several Bootstraps read is rotated once, and
        norm2_linprod = the reference's statistic (`stats`) with the output of a shared rotation weighted by its table's
        `table_fusion_factor` instead of 1 -- what the parameter choice of a fused evaluation has to carry."""
        readers = {}
        for instr in self.instructions:
            if isinstance(instr, LutExecEnv.Bootstrap):
                readers[instr.val.name] = readers.get(instr.val.name, 0) + 1
        norm2, rotations = {}, 0
        for instr in self.instructions:
            if isinstance(instr, LutExecEnv.Input):
                norm2[instr.name] = 1
            elif isinstance(instr, LutExecEnv.LinearProd):
                norm2[instr.name] = sum(c * c * norm2[v.name] for c, v in instr.coef_vals)
            else:
                shared = readers[instr.val.name] >= 2
                norm2[instr.name] = table_fusion_factor(instr.table, p) if shared else 1
        rotations = sum(1 for n in readers.values() if n >= 2) + sum(n for n in readers.values() if n == 1)
        return dict(nb_rotation=rotations, norm2_linprod=max(norm2.values(), default=1))

    def tables(self):
        return [i.table for i in self.instructions if isinstance(i, LutExecEnv.Bootstrap)]

    def schedule(self):
        """Bootstrap levels: level(input)=0, level(LinearProd)=max over its sources,
        level(Bootstrap)=level(source)+1.  Returns (levels: name->int, depth, widths per level)."""
        level = {"0": 0, "1": 0}
        widths = {}
        for instr in self.instructions:
            if isinstance(instr, LutExecEnv.Input):
                level[instr.name] = 0
            elif isinstance(instr, LutExecEnv.LinearProd):
                level[instr.name] = max((level[v.name] for _, v in instr.coef_vals), default=0)
            else:
                lv = level[instr.val.name] + 1
                level[instr.name] = lv
                widths[lv] = widths.get(lv, 0) + 1
        depth = max(widths, default=0)
        return level, depth, [widths.get(l, 0) for l in range(1, depth + 1)]

    # ---- lowering to the C ABI's flat program ------------------------------------------------------
    def lower(self):
        """Flat arrays of `fbs_program_desc` (include/fbs_exec.h) plus the distinct tables."""
        if self._lowered is not None:
            return self._lowered
        inputs = [i for i in self.instructions if isinstance(i, LutExecEnv.Input)]
        body = [i for i in self.instructions if not isinstance(i, LutExecEnv.Input)]
        wire = {inp.name: k for k, inp in enumerate(inputs)}
        for k, instr in enumerate(body):
            wire[instr.name] = len(inputs) + k
        kind, arg0, arg1, const_coef, term_coef, term_src = [], [], [], [], [], []
        tables, table_id = [], {}
        for instr in body:
            if isinstance(instr, LutExecEnv.LinearProd):
                kind.append(0)
                arg0.append(len(term_src))
                arg1.append(len(instr.coef_vals))
                const = int(instr.const_coef)
                for c, v in instr.coef_vals:
                    if isinstance(v, LutExecEnv.Const):     # only reachable through hand-made nodes
                        const += int(c) * int(v.value)
                        arg1[-1] -= 1
                        continue
                    term_coef.append(int(c))
                    term_src.append(wire[v.name])
                const_coef.append(const)
            else:
                key = tuple(int(t) for t in instr.table)
                if key not in table_id:
                    table_id[key] = len(tables)
                    tables.append(list(key))
                kind.append(1)
                arg0.append(wire[instr.val.name])
                arg1.append(table_id[key])
                const_coef.append(0)
        out_names, out_wire = [], []
        for name, node in self.outputs.items():
            out_names.append(name)
            out_wire.append(-1 - int(node.value) if isinstance(node, LutExecEnv.Const) else wire[node.name])
        self._lowered = dict(input_names=[i.name for i in inputs], kind=kind, arg0=arg0, arg1=arg1,
                             const_coef=const_coef, term_coef=term_coef, term_src=term_src, tables=tables,
                             out_names=out_names, out_wire=out_wire)
        return self._lowered

    # ---- the hot path ------------------------------------------------------------------------------
    def eval(self, input_values, config: ExecConfig | None = None):
        """Reference contract (:208-229): {input name: array-like of bits} -> {output name: np.ndarray
        of ints}; a constant output comes back as a python int.  Evaluated on ciphertexts on the GPU."""
        from . import _native as nat
        cfg = config or self.exec_config
        low = self.lower()
        p = cfg.fbs_size or min_fbs_size(low["tables"])
        for t in low["tables"]:
            assert table_is_valid(t, p), "table %s cannot be evaluated by one bootstrap at fbs_size %d" % (t, p)
        names = low["input_names"]
        cols = [np.asarray(input_values[n]).reshape(-1) for n in names]
        T = max((len(c) for c in cols), default=1)
        ctx, fuse = cfg.choose(self, p, samples=T)

        bits = np.stack([np.broadcast_to(c, (T,)) for c in cols]).astype(np.int64) if cols else np.zeros((0, T), np.int64)
        assert bits.size == 0 or (bits.min() >= 0 and bits.max() <= 1), "inputs are bits"

        program = cfg.program_for(ctx, low, fuse)
        cts = ctx.encrypt(bits, nonce0=cfg.take_nonces(bits.size))
        out = ctx.decrypt(program.eval(cts, T))
        result = {}
        for k, name in enumerate(low["out_names"]):
            w = low["out_wire"][k]
            result[name] = (-1 - w) if w < 0 else out[k].astype(int)
        return result


FbsExecEnv = LutExecEnv


# --------------------------------------------------------------------------------------------
# readers for the two text formats the reference emits (it has no reader of its own)
# --------------------------------------------------------------------------------------------
_TERM = re.compile(r"^(-?\d+) \* (\S+)$")
_TABLE_ENTRY = re.compile(r"^(?:np\.int64\((-?\d+)\)|(-?\d+))$")


class _Loader:
    """Builds a LutExecEnv with the names given in the text (ids may have gaps: CSE consumed them)."""

    def __init__(self, merge_linear_prods=True):
        self.env = LutExecEnv(merge_linear_prods)
        self.nodes = {}

    def ref(self, name):
        node = self.nodes.get(name)
        if node is None:
            if name in ("0", "1"):
                return self.env.const(int(name))
            node = self.nodes[name] = self.env.input(name)      # undefined name = primary input
        return node

    def define(self, node):
        assert node.name not in self.nodes, "wire %s defined twice" % node.name
        got = self.env._add_instr(node)
        self.nodes[node.name] = got
        m = re.fullmatch(r"m(\d+)", node.name)
        if m:
            self.env._unique_id = max(self.env._unique_id, int(m.group(1)))
        return got


def parse_fbs(text, inputs=None, merge_linear_prods=True):
    """Read what `LutExecEnv.print(show_outputs=True)` wrote.  `inputs` optionally fixes the order of
    the primary inputs (the printer omits them by default)."""
    ld = _Loader(merge_linear_prods)
    for name in inputs or ():
        ld.ref(name)
    for raw in text.splitlines():
        line = raw.strip()
        if not line:
            continue
        if line.startswith("Output "):
            name, _, target = line[len("Output "):].partition(" = ")
            ld.env.output(name, ld.ref(target.strip()))
            continue
        name, _, rhs = line.partition(" = ")
        rhs = rhs.strip()
        if rhs.startswith("Input("):
            ld.ref(name)
        elif rhs.startswith("Bootstrap("):
            src, _, tab = rhs[len("Bootstrap("):-1].partition(", ")
            entries = []
            for e in tab.strip()[1:-1].split(","):
                m = _TABLE_ENTRY.match(e.strip())
                assert m, "bad table entry %r" % e
                # numpy >= 2 leaks `np.int64(1)` reprs into the reference's tables; keep the flavour so
                # that printing the program again gives the same text
                entries.append(np.int64(m.group(1)) if m.group(1) is not None else int(m.group(2)))
            src_node = ld.ref(src)
            assert len(entries) == ld.env.max_val[src_node.name] + 1, "table length does not match %s" % src
            ld.define(LutExecEnv.Bootstrap(name, src_node, entries))
        else:
            terms, const = [], 0
            for piece in rhs.split(" + "):
                piece = piece.strip()
                m = _TERM.match(piece)
                if m:
                    terms.append((int(m.group(1)), ld.ref(m.group(2))))
                else:
                    const += int(piece)
            ld.define(LutExecEnv.LinearProd(name, terms, const))
    return ld.env


def parse_lbf(text, merge_linear_prods=True):
    """Read what `LutExecEnv.write_lbf` wrote."""
    logical, pending = [], ""
    for raw in text.splitlines():
        if raw.rstrip().endswith("\\"):
            pending += raw.rstrip()[:-1] + " "
            continue
        logical.append((pending + raw).strip())
        pending = ""
    ld = _Loader(merge_linear_prods)
    outputs = []
    k = 0
    while k < len(logical):
        words = logical[k].split()
        k += 1
        if not words:
            continue
        if words[0] == ".inputs":
            for name in words[1:]:
                ld.ref(name)
        elif words[0] == ".outputs":
            outputs = words[1:]
        elif words[0] == ".lincomb":
            *srcs, name = words[1:]
            nums = [int(w) for w in logical[k].split()]
            k += 1
            if name in outputs and name not in ld.nodes and len(srcs) == 1 and nums == [1] and not re.fullmatch(r"m\d+", name):
                ld.env.output(name, ld.ref(srcs[0]))            # the trailing "output = 1 * node" records
                continue
            if name in outputs and name in ld.nodes and len(srcs) == 1 and nums == [1]:
                ld.env.output(name, ld.ref(srcs[0]))
                continue
            coefs, const = nums[:len(srcs)], (nums[len(srcs)] if len(nums) > len(srcs) else 0)
            ld.define(LutExecEnv.LinearProd(name, [(c, ld.ref(s)) for c, s in zip(coefs, srcs)], const))
        elif words[0] == ".bootstrap":
            src, name = words[1], words[2]
            entries = [int(ch) for ch in logical[k].strip()]
            k += 1
            ld.define(LutExecEnv.Bootstrap(name, ld.ref(src), entries))
        else:
            raise ValueError("unknown .lbf record: %s" % logical[k - 1])
    return ld.env
