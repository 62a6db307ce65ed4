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
from tfhe_fbs_map_amd.distributed import choose_sharding, launch_family, launch_ms, plan_levels, split_range

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


@pytest.mark.parametrize("name,T", [("full_adder__search_p7", 5), ("adder8__search_p7", 3), ("edge_outputs", 4),
                                    ("full_adder__search_p7", 1)])        # T = 1: levels narrower than the world, empty slices
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


def test_four_ranks_as_two_sample_groups_of_two_gate_ranks(tmp_path):
    """The 2-D layout `choose_sharding` can ask for: ranks 0, 1 cut the levels of samples [0, 3), ranks 2, 3 those of samples
    [3, 5); one all-gather per level inside each pair, one gather of the outputs over all four at the end."""
    name, T = "full_adder__search_p7", 5
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="4", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), name, str(T), out],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(4)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    z = np.load(out)
    rec = load_fixture(name)
    env_ = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env_.lower()
    _, depth, widths = env_.schedule()
    assert int(z["world"]) == 4
    for k, w in enumerate(low["out_wire"]):
        if w >= 0:
            for mode in ("gate", "sample", "grid"):
                assert np.array_equal(z[mode][k], z["ref"][k]), mode
    assert int(z["grid_collectives"]) == depth + 1
    assert int(z["grid_fbs"]) <= -(-sum(widths) * 3 // 2) + depth          # rank 0: half of the levels of three samples


def test_choose_sharding_prefers_samples_and_falls_back_to_gates():
    """Samples are cut first (no collectives); gate groups appear when there are fewer samples than ranks; the prediction is
    the launch-time staircase summed over the levels."""
    widths = [3, 1, 2, 13, 8]
    c = choose_sharding(widths, 1000, 8)
    assert (c["sample_groups"], c["gate_groups"]) == (8, 1)
    assert abs(c["predicted_ms"] - sum(launch_ms(w * 125) for w in widths)) < 1e-9
    assert c["single_gpu_ms"] > c["predicted_ms"] and 1.0 < c["predicted_speedup"] <= 8.0
    c = choose_sharding(widths, 2, 8)
    assert (c["sample_groups"], c["gate_groups"]) == (2, 4)
    c = choose_sharding([400, 300], 1, 4)
    assert (c["sample_groups"], c["gate_groups"]) == (1, 4) and c["predicted_speedup"] > 1.5
    assert {(k["sample_groups"], k["gate_groups"]) for k in choose_sharding(widths, 64, 8)["candidates"]} == {(1, 8), (2, 4), (4, 2), (8, 1)}
    # the staircase: one more bootstrap than a CU-round costs a second round; whole rounds of 1024 add up
    assert launch_ms(257) > 1.7 * launch_ms(256) and launch_ms(4096) < 4.2 * launch_ms(1024)


def test_layouts_are_priced_on_the_staircase_of_the_set_actually_loaded():
    """`choose_sharding(..., params=)`: every parameter set follows the launch-time staircase of the kernels IT runs on
    (schedule.LAUNCH_FAMILIES), scaled by its blind-rotation steps -- not the benchmark shape's times a scalar."""
    from tfhe_fbs_map_amd import P1024
    from tfhe_fbs_map_amd.params import choose_params
    k2, n2048, p31 = choose_params(15, 70, glwe_dims=(1, 2)), choose_params(15, 70), choose_params(31, 325)
    assert launch_family(k2) == ("k2", 1.0) and launch_family(n2048)[0] == "n2048" and launch_family(p31)[0] == "n2048_l2"
    assert launch_family(P1024) == ("p1024", 1.0)
    assert launch_family(choose_params(4, 2, glwe_dims=(1, 2))) == ("k2", (630 // 2) / 367)      # fewer steps, the same kernels
    name, scale = launch_family(choose_params(4, 2))                                             # no staircase of its own: P1024's
    assert name == "p1024" and 0.5 < scale < 1.0
    # the k = 2 family: one bootstrap per CU costs one round of the twelve-wave shape, 257 two, 769 a round of four per workgroup
    assert launch_ms(256, params=k2) < 2.5 < 4.0 < launch_ms(257, params=k2) < 4.6 and launch_ms(769, params=k2) < 7.0
    assert launch_ms(2048, params=k2) < 2.05 * launch_ms(1024, params=k2)
    assert launch_ms(1124, params=k2) < launch_ms(1024, params=k2) + launch_ms(100, params=k2) + 1e-9
    assert launch_ms(256, params=k2) < launch_ms(256, params=n2048) < launch_ms(256, params=p31)
    # an adder's levels (widths 1-3) at the harness's T = 1000 over 8 ranks: priced per set, and the ciphertexts of a k = 2 set are 2 N + 1 words
    widths = [1, 2, 3, 2] * 8
    a, b = choose_sharding(widths, 1000, 8, params=k2), choose_sharding(widths, 1000, 8, params=p31)
    assert a["predicted_ms"] < b["predicted_ms"] and a["single_gpu_ms"] < b["single_gpu_ms"]
    assert 5.0 < a["predicted_speedup"] <= 8.0
    gate = [c for c in choose_sharding([300, 400], 4, 8, params=k2)["candidates"] if c["gate_groups"] == 8][0]
    assert gate["allgather_ms"] > 0
    # the k = 3, N = 512 family (the default sets for p <= 8): a round of the chip is 768 bootstraps, one bootstrap per workgroup up to one
    # per CU, two up to two; a launch of a round + 256 is cut and costs the two launches; ahead of the k = 2 set at every size
    k3, k2p4 = choose_params(4, 2, glwe_dims=(1, 2, 3)), choose_params(4, 2, glwe_dims=(1, 2))
    assert launch_family(k3) == ("k3", 1.0) and launch_family(choose_params(7, 10, glwe_dims=(1, 2, 3)))[0] == "k3"
    assert launch_ms(256, params=k3) < 1.9 < 2.4 < launch_ms(257, params=k3) < launch_ms(512, params=k3) < 3.0 < launch_ms(768, params=k3) < 4.2
    assert abs(launch_ms(1024, params=k3) - (launch_ms(768, params=k3) + launch_ms(256, params=k3))) < 0.2
    assert launch_ms(3072, params=k3) < 4.05 * launch_ms(768, params=k3)
    for B in (1, 64, 256, 512, 768, 1024, 1536, 3072, 6144):
        assert launch_ms(B, params=k3) < launch_ms(B, params=k2p4), B
