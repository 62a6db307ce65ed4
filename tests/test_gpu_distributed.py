"""The N > 1 data path on real kernels: two ranks, each with its own context and device-resident wires, share the test
box's one GPU and exchange over gloo (RCCL refuses two ranks on one device; with more GPUs the same runners use it,
tests/test_gpu_levels.py covers that code path with one rank).  Gate-sharded and sample-sharded results must be the
single-process result, bit for bit."""
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


@pytest.mark.parametrize("name,T,flavour", [("adder8__search_p7", 5, ""), ("adder8__basic_p2", 3, ""), ("edge_outputs", 4, ""),
                                            ("full_adder__search_p7", 1, ""),   # T = 1: levels narrower than the world, empty slices
                                            ("adder8__search_p7", 5, "k2"), ("edge_outputs", 4, "k2"),    # GLWE dimension 2
                                            ("adder8__search_p7", 5, "k3"), ("edge_outputs", 4, "k3")])   # ... and 3 at N = 512 (the default for p <= 8)
def test_two_ranks_on_one_gpu_bit_identical(tmp_path, name, T, flavour):
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), name, str(T), out] + ([flavour] if flavour else []),
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    z = np.load(out)
    rec = load_fixture(name)
    _, expect = subsample(rec, T)
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    assert int(z["world"]) == 2
    for k, w in enumerate(low["out_wire"]):
        if w < 0:
            continue
        assert np.array_equal(z["gate"][k], z["ref"][k])
        assert np.array_equal(z["sample"][k], z["ref"][k])
        assert np.array_equal(z["dec"][k], expect[low["out_names"][k]])
    assert int(z["gate_collectives"]) == int(z["depth"]) and int(z["sample_collectives"]) == 1
    total = int(z["n_bootstrap"]) * T
    assert total // 2 - int(z["depth"]) <= int(z["gate_fbs"]) <= -(-total // 2) + int(z["depth"])      # rank 0 did half


@pytest.mark.parametrize("flavour", ["fused", "fused_k2", "fused_k3"])
def test_two_ranks_on_a_fused_program(tmp_path, flavour):
    """A program loaded with FBS_LOAD_FUSE_TABLES (several tables on one blind rotation), cut across two ranks both ways.  Gate-
    sharded, the unit dealt out is the ROTATION: rows are (k + 1) N words, a shared rotation's accumulator travels in its row, and
    every rank cuts the tables out of the gathered accumulators.  Both layouts return the single-process (fused) ciphertexts, and
    rank 0 did about half of the rotations.  (fused_k2: the same at GLWE dimension 2.)"""
    name, T = "adder8__basic_p2", 4
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), name, str(T), out, flavour],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    z = np.load(out)
    rec = load_fixture(name)
    _, expect = subsample(rec, T)
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    assert int(z["gate_collectives"]) == int(z["depth"]) and int(z["sample_collectives"]) == 1
    for k, w in enumerate(low["out_wire"]):
        if w >= 0:
            assert np.array_equal(z["gate"][k], z["ref"][k])
            assert np.array_equal(z["sample"][k], z["ref"][k])
            assert np.array_equal(z["dec_sample"][k], expect[low["out_names"][k]])
    assert int(z["gate_fbs"]) < int(z["n_bootstrap"]) * T * 0.6          # rotations, not tables: 22 of them for 37 tables
