"""RCCL with two or more ranks, one GPU each: the deployment shape of north_star's multi-GPU path.  Skipped with its reason on a
one-GPU box (the boxes this build is developed on); on a node with several MI355X it runs by itself, no edit needed:
every layout of distributed.ShardedRunner -- gate-sharded (one all-gather per level), sample-sharded, the 2 x 2 grid from four
GPUs on, a program with shared rotations and a k = 2 context -- must return, bit for bit, what one process returns; and
`bench.py --gpus 2` must report its sharded legs on the nccl backend with one collective per level."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import load_fixture, subsample
from tfhe_fbs_map_amd import parse_fbs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_count():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU in this process)


def world_sizes():
    n = gpu_count()
    return [w for w in (2, 4, 8) if w <= n]


needs_two = pytest.mark.skipif(gpu_count() < 2, reason="needs >= 2 GPUs on the node: RCCL refuses two ranks on one device "
                               "(tests/test_gpu_distributed.py runs the same data path with two ranks sharing the GPU over gloo)")


def launch(world, script_args, timeout=900):
    """torchrun as a child process, started before this process makes any GPU call of its own in the test"""
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="2")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@needs_two
@pytest.mark.parametrize("name,T,flavour", [("adder8__search_p7", 5, ""), ("edge_outputs", 4, ""), ("full_adder__search_p7", 1, ""),
                                            ("adder8__basic_p2", 4, "fused"), ("adder8__search_p7", 5, "k2"), ("adder8__search_p7", 5, "k3")])
def test_sharded_runners_on_rccl_bit_identical(tmp_path, name, T, flavour):
    for world in world_sizes():
        out = str(tmp_path / ("res%d.npz" % world))
        rc = launch(world, [os.path.join(ROOT, "tests", "dist_multi_worker.py"), name, str(T), out] + ([flavour] if flavour else []))
        assert rc.returncode == 0, rc.stderr[-3000:]
        z = np.load(out)
        assert str(z["backend"]) == "nccl" and int(z["world"]) == world
        rec = load_fixture(name)
        _, expect = subsample(rec, T)
        low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
        layouts = ["gate", "sample"] + (["grid"] if world >= 4 else [])
        for k, w in enumerate(low["out_wire"]):
            if w < 0:
                continue
            for lay in layouts:
                assert np.array_equal(z[lay][k], z["ref"][k]), (world, lay, low["out_names"][k])
            assert np.array_equal(z["dec"][k], expect[low["out_names"][k]])
        assert int(z["gate_collectives"]) == int(z["depth"]) and int(z["sample_collectives"]) == 1
        if world >= 4:
            assert int(z["grid_collectives"]) <= int(z["depth"]) + 1           # per level inside a gate group (levels with an empty batch skip theirs), and the outputs once
        units = int(z["n_rotations"] if flavour == "fused" else z["n_bootstrap"]) * T
        assert units // world - int(z["depth"]) <= int(z["gate_fbs"]) <= -(-units // world) + int(z["depth"])     # rank 0 did its share


@needs_two
def test_bench_two_gpus_reports_rccl_legs():
    """`python bench.py --gpus 2 --steps 2` exactly as the driver calls it (no WORLD_SIZE: bench.py starts its own ranks)."""
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                         "--sharded-circuit", "adder8__search_p15", "--sharded-samples", "16"],
                        capture_output=True, text=True, timeout=1200,
                        env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")})
    assert rc.returncode == 0, rc.stderr[-3000:]
    d = json.loads([ln for ln in rc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["decrypt_ok"]
    legs = d["sharded"]["legs"]
    rec = load_fixture("adder8__search_p15")
    from tfhe_fbs_map_amd.schedule import plan_levels
    depth = plan_levels(parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower())["depth"]
    for name, leg in legs.items():
        assert leg["backend"] == "nccl" and leg["rccl_ranks"] == 2 and leg["decrypt_ok"], name
    assert legs["gate"]["collectives_per_step"] == depth and legs["sample"]["collectives_per_step"] == 0
    assert d["sharded"]["params"]["security_bits_estimate"] >= 127.9


@pytest.mark.parametrize("flavour", ["", "fused", "k2", "k3"])
def test_the_worker_itself_on_one_rank(tmp_path, flavour):
    """The worker of the tests above with world size 1 on the nccl backend -- on ANY GPU box, the one-GPU ones included: the file
    that waits for a multi-GPU node is at least run end to end (process group, layouts, the npz it hands back) where this build
    can run it."""
    out = str(tmp_path / "res1.npz")
    name, T = ("adder8__basic_p2", 4) if flavour == "fused" else ("adder8__search_p7", 5)
    rc = launch(1, [os.path.join(ROOT, "tests", "dist_multi_worker.py"), name, str(T), out] + ([flavour] if flavour else []))
    assert rc.returncode == 0, rc.stderr[-3000:]
    z = np.load(out)
    assert str(z["backend"]) == "nccl" and int(z["world"]) == 1
    rec = load_fixture(name)
    _, expect = subsample(rec, T)
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    for k, w in enumerate(low["out_wire"]):
        if w >= 0:
            assert np.array_equal(z["gate"][k], z["ref"][k]) and np.array_equal(z["sample"][k], z["ref"][k])
            assert np.array_equal(z["dec"][k], expect[low["out_names"][k]])


def test_skips_say_why_on_a_one_gpu_box():
    """(so that a one-GPU run shows this module was considered, and what it would take to run it)"""
    if gpu_count() >= 2:
        pytest.skip("several GPUs: the tests above run")
    assert needs_two.kwargs["reason"].startswith("needs >= 2 GPUs")
