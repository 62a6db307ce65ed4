"""N > 1 path on CPU: world_size-2 gloo ranks run tfhe_fbs_map_amd.distributed's two runners (gate-sharded with
one all-gather per level, sample-sharded with none) and must reproduce the single-process result bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import load_fixture, subsample
from tfhe_fbs_map_amd import parse_fbs
from tfhe_fbs_map_amd.distributed import plan_levels, split_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_covers_exactly():
    for total, parts in ((10, 3), (7, 8), (64, 2), (1, 2), (0, 2)):
        got = []
        for r in range(parts):
            a, b, _ = split_range(total, parts, r)
            got += list(range(a, b))
        assert got == list(range(total))


def test_plan_matches_facade_schedule():
    rec = load_fixture("mul4__search_p15")
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    plan = plan_levels(env.lower())
    _, depth, widths = env.schedule()
    assert plan["depth"] == depth and [len(b["src"]) for b in plan["boot"]] == widths
    done = set(range(plan["n_inputs"]))
    for L in range(depth + 1):                 # every stage reads only wires produced earlier
        for st in plan["lin"][L]:
            assert set(st["srcs"]) <= done
            done |= set(st["dst"])
        if L < depth:
            assert set(plan["boot"][L]["src"]) <= done
            done |= set(plan["boot"][L]["dst"])


@pytest.mark.parametrize("name,T", [("full_adder__search_p7", 5), ("adder8__search_p7", 3), ("edge_outputs", 4)])
def test_two_ranks_bit_identical(tmp_path, name, T):
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), name, str(T), out],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    z = np.load(out)
    rec = load_fixture(name)
    _, expect = subsample(rec, T)
    env_ = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env_.lower()
    wire_outs = [k for k, w in enumerate(low["out_wire"]) if w >= 0]
    assert int(z["world"]) == 2
    for k in wire_outs:
        assert np.array_equal(z["gate"][k], z["ref"][k])          # gate-sharded == single process, every word
        assert np.array_equal(z["sample"][k], z["ref"][k])        # sample-sharded too
        assert np.array_equal(z["dec"][k], expect[low["out_names"][k]])
    _, depth, widths = env_.schedule()
    assert int(z["gate_collectives"]) == depth                    # one all-gather per level
    assert int(z["sample_collectives"]) == 1                      # only the final gather of outputs
    total = sum(widths) * T
    assert int(z["gate_fbs"]) <= -(-total // 2) + depth * 1 and int(z["gate_fbs"]) >= total // 2 - depth
