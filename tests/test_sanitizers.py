"""AddressSanitizer + UndefinedBehaviorSanitizer on everything of this path that is HOST code (the GPU pool has
no GPU sanitizer, so this is where they run -- in the CPU suite, every round):

* the PRODUCT's program loader arithmetic -- levels, wire slots handed out by liveness, the stage tables the level kernels index,
  shared rotations -- factored out of fbs_program_load_ex into csrc/fbs_plan.cpp so that it runs without a device: planned and then
  EXECUTED in the clear on wire slots, for all the reference-mapped fixtures and for random programs, against the reference's
  own outputs;
* the product's key generation / encryption / decryption / test vectors (csrc/fbs_host.cpp);
* the oracle and the tuned CPU baseline (test infrastructure) at toy parameter sets.
tests/c/Makefile builds the two harnesses with -fsanitize=address,undefined; a finding aborts the harness and fails the test."""
import io
import os
import subprocess

import numpy as np
import pytest

from oracle import lut_oracle
from tests.helpers import fixture_names, load_fixture, subsample
from tfhe_fbs_map_amd import parse_fbs
from tfhe_fbs_map_amd.schedule import plan_levels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "c", "build")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2")


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), "asan"])
    return os.path.join(BUILD, "host_harness"), os.path.join(BUILD, "oracle_harness")


def describe(low, ins, T, fusable=None):
    """the text the harness reads (tests/c/host_harness.cpp)"""
    tables = low["tables"]
    n_instr = len(low["kind"])
    parts = [[len(low["input_names"]), n_instr, len(low["term_src"]), len(low["out_wire"]), len(tables), T],
             low["kind"], low["arg0"], low["arg1"], low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"]]
    for t in tables:
        parts.append([len(t)] + [int(v) for v in t])
    parts.append([1] * len(tables) if fusable is None else fusable)
    for name in low["input_names"]:
        parts.append([int(v) for v in np.asarray(ins[name]).reshape(-1)[:T]])
    return "\n".join(" ".join(str(int(v)) for v in p) for p in parts) + "\n"


def run_plan(exe, text):
    r = subprocess.run([exe, "plan"], input=text, capture_output=True, text=True, env=ENV, timeout=120)
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stdout[-500:] + r.stderr[-3000:]
    return r.stdout.splitlines()


def check_program(exe, low, ins, expect, T):
    lines = run_plan(exe, describe(low, ins, T))
    stats = {ln.split()[0]: dict(zip(ln.split()[1::2], (int(v) for v in ln.split()[2::2]))) for ln in lines if ln.startswith(("plain", "fused"))}
    widths = [[int(v) for v in ln.split()[1:]] for ln in lines if ln.startswith("widths")]
    ref = plan_levels(low)                                   # the Python restatement of the same rule (schedule.py)
    assert stats["plain"]["depth"] == ref["depth"] and widths[0] == [len(b["dst"]) for b in ref["boot"]]
    n_boot = sum(1 for k in low["kind"] if k == 1)
    assert stats["plain"]["bootstraps"] == stats["plain"]["rotations"] == n_boot
    assert stats["plain"]["keyswitches"] == sum(len(set(b["src"])) for b in ref["boot"]) <= n_boot
    assert stats["plain"]["slots"] <= stats["plain"]["wires"] and stats["fused"]["rotations"] <= n_boot
    out = [int(v) for v in [ln for ln in lines if ln.startswith("outputs")][0].split()[1:]]
    out = np.array(out, np.int64).reshape(len(low["out_wire"]), T)
    for k, name in enumerate(low["out_names"]):
        e = expect[name]
        assert np.array_equal(out[k], np.full(T, e) if np.ndim(e) == 0 else np.asarray(e)[:T]), name
    return stats


def test_crypto_host_code_and_the_oracle_under_the_sanitizers(harness):
    host, orc = harness
    for cmd in ([host, "crypto"], [orc]):
        r = subprocess.run(cmd, capture_output=True, text=True, env=ENV, timeout=600)
        assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stdout[-500:] + r.stderr[-3000:]
        assert r.stdout.count(" ok") >= 3


def test_the_loader_plan_on_every_fixture_under_the_sanitizers(harness):
    """All reference-mapped programs of tests/golden (the builder's edge cases included): planned, executed in the clear slot by
    slot -- with one rotation per table and with shared rotations -- equal to the reference's own eval outputs."""
    host, _ = harness
    names = fixture_names()
    assert len(names) >= 200
    shared = 0
    for name in names:
        rec = load_fixture(name)
        T = 4 if rec["stats"]["nb_bootstrap"] > 2000 else 8
        low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"], merge_linear_prods=name != "edge_nomerge").lower()
        ins, expect = subsample(rec, T)
        T = min(T, len(next(iter(ins.values()))) if ins else T)
        stats = check_program(host, low, ins, expect, T)
        shared += stats["fused"]["rotations"] < stats["plain"]["rotations"]
    assert shared >= 3                                         # adder8 / 2_input_gates / half_adder under `basic` share sources


@pytest.mark.parametrize("seed", range(10))
def test_the_loader_plan_on_random_programs_under_the_sanitizers(harness, seed):
    """Shapes the mappers never emit (tests/test_gpu_random_programs.py's generators: multi-valued tables, negative coefficients,
    lincomb outputs, deep chains, several tables per source)."""
    from tests.test_gpu_random_programs import random_program, random_shared_program
    host, _ = harness
    for env, _ in (random_program(seed), random_shared_program(seed)):
        rng = np.random.default_rng(3000 + seed)
        names = [i.name for i in env.instructions if isinstance(i, type(env).Input)]
        ins = {n: rng.integers(0, 2, 12) for n in names}
        buf = io.StringIO()
        env.print(os=buf, show_outputs=True)
        expect = lut_oracle.eval_fbs_text(buf.getvalue(), ins)
        low = env.lower()
        check_program(host, low, ins, {k: (int(v) if np.ndim(v) == 0 else np.asarray(v)) for k, v in expect.items()}, 12)


def test_malformed_descriptions_are_refused_not_crashed_on(harness):
    host, _ = harness
    rec = load_fixture("full_adder__search_p7")
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    ins, _ = subsample(rec, 2)

    def broken(**kw):
        bad = {k: (list(v) if isinstance(v, (list, tuple, np.ndarray)) else v) for k, v in low.items()}
        for k, f in kw.items():
            bad[k] = f(bad[k])
        return run_plan(host, describe(bad, ins, 2))

    first_boot = list(low["kind"]).index(1)
    first_lin = list(low["kind"]).index(0)
    cases = [dict(arg1=lambda a: [99 if i == first_boot else v for i, v in enumerate(a)]),              # table id out of range
             dict(arg0=lambda a: [10 ** 6 if i == first_boot else v for i, v in enumerate(a)]),         # a bootstrap reads a later wire
             dict(arg1=lambda a: [10 ** 6 if i == first_lin else v for i, v in enumerate(a)]),          # term range out of bounds
             dict(term_src=lambda a: [10 ** 6] + list(a[1:])),                                         # a lincomb reads a later wire
             dict(kind=lambda a: [7] + list(a[1:])),                                                   # unknown instruction kind
             dict(out_wire=lambda a: [10 ** 6] + list(a[1:]))]                                         # output wire out of range
    for case in cases:
        lines = broken(**case)
        assert lines and lines[0].startswith("error -1 "), (case, lines)
