"""bench.py is what the driver runs: its JSON line must keep the contract fields, and the circuit workload must agree with
the library it drives.  Small sizes here; the real sizes are the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900)
    assert rc.returncode == 0, rc.stderr[-2000:]
    lines = [ln for ln in rc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    return json.loads(lines[0])


def test_headline_line_has_the_contract_fields():
    d = run_bench("--steps", "3", "--warmup", "1", "--batch", "256", "--cpu-sample", "8")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "secure"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "FBS/s" and d["vs_baseline"] is None and d["decrypt_ok"]
    assert abs(d["value"] - 256 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "fp64_valu" and r["kernel"].startswith("k_blind_rotate") and r["avg_launch_ms"] > 0
    assert 0 < r["frac"] == r["algorithmic_frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-12
    assert 0 < r["key_stream_vs_hbm_peak"] and "hbm_algorithmic_frac" not in r
    assert r["launches"][0]["kernel"] == r["kernel"] and r["launches"][0]["launches_per_step"] == 1
    # executed-instruction figures come from an offline PMC record and only if it was taken on these very kernel sources
    assert (r["valu_frac"] is None and ("stale" in r["pmc"] or "no record" in r["pmc"] or "record is for" in r["pmc"])) or \
        (0 < r["valu_frac"] < 1 and "profiles/" in r["pmc"])
    assert "test-grade" in d["config"]["params"]["randomness"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["bit_exact_vs_gpu"] and c["cores"] >= 1 and c["cpu_model"] and c["one_thread"]["value"] > 0
    p = d["config"]["params"]
    assert p["sigma_lwe"] == 64 and p["security_bits_estimate"] < 60          # the headline says what it is
    s = d["secure"]
    assert s["decrypt_ok"] and s["params"]["security_bits_estimate"] >= 127.9 and s["margin_sigmas_at_norm2_70"] >= 6.0
    assert s["n1024_p4"]["decrypt_ok"] and s["n1024_p4"]["params"]["N"] == 1024
    assert s["p31"]["decrypt_ok"] and s["p31"]["params"]["security_bits_estimate"] >= 127.9 and s["p31"]["params"]["p"] == 31
    # every 128-bit leg holds a sample of its TIMED batch to the scalar oracle, word for word; at batch 256 the default
    # set runs on the twelve-wave latency shape; and the deployable number sits at the top level beside the benchmark shape
    legs = [s, s["k1"], s["one_key_bit_per_step"], s["n1024_p4"], s["n1024_p4"]["k2"], s["p31"], s["p31"]["one_key_bit_per_step"]]
    if "k3_n512" in s["n1024_p4"]:                         # (timed on batches of a round or more only)
        legs.append(s["n1024_p4"]["k3_n512"])
    assert all(leg["bit_exact_vs_oracle"] is True and "scalar oracle" in leg["oracle_sample"] for leg in legs)
    assert s["params"]["k"] == 2 and s["blind_rotate_kernel"] == "k_blind_rotate_cu_k2" and s["k1"]["params"]["k"] == 1
    v = d["value_secure"]
    assert v["value"] == s["value"] and v["bit_exact_vs_oracle"] and v["params"]["security_bits_estimate"] >= 127.9 and v["params"]["k"] == 2
    f = d["shared_rotations"]                                                # several tables on one blind rotation
    assert f["plain"]["all_sums_correct"] and f["fused"]["all_sums_correct"]
    assert f["fused"]["blind_rotations"] < f["fused"]["tables"] == f["plain"]["tables"] == f["plain"]["blind_rotations"]
    assert f["fused"]["params"]["security_bits_estimate"] >= 127.9 and f["fused"]["margin_sigmas_at_its_norm2"] >= 5.9


@pytest.mark.parametrize("mode", ["gate", "sample"])
def test_circuit_workload(mode):
    d = run_bench("--workload", "circuit", "--mode", mode, "--circuit", "adder8__search_p15", "--samples", "8",
                  "--steps", "1", "--warmup", "1")
    assert d["decrypt_ok"] and d["config"]["mode"] == mode and d["config"]["rccl_ranks"] == 1
    assert d["scaling"] == ("strong" if mode == "gate" else "weak")
    assert d["config"]["wire_slots"] < d["config"]["wires"]
    assert d["value"] > 0 and set(d["kernels_ms_per_step"]) == {"keyswitch", "blind_rotate", "lincomb"}


def test_circuit_workload_at_the_secure_set():
    d = run_bench("--workload", "circuit", "--mode", "gate", "--circuit", "adder8__search_p15", "--samples", "8",
                  "--steps", "1", "--warmup", "1", "--secure")
    assert d["decrypt_ok"] and d["config"]["params"]["security_bits_estimate"] >= 127.9
    assert "128-bit" in d["config"]["workload"]


def test_two_rank_line_carries_the_sharded_legs():
    """`bench.py --gpus 2` as the driver's torchrun starts it, on THIS box's one GPU (test hook FBS_BENCH_SHARE_GPU: the two ranks
    share the device and meet over gloo, RCCL refusing two ranks on one device): the weak-scaling headline with its max-over-ranks
    timing, and nested in the same line one circuit cut three ways with ranks, collectives per step, all-gather time, kernel
    instantiations and a decrypt check per layout."""
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "128",
           "--sharded-circuit", "adder8__search_p15", "--sharded-samples", "6"]
    rc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, FBS_BENCH_SHARE_GPU="1"))
    assert rc.returncode == 0, rc.stderr[-3000:]
    lines = [ln for ln in rc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["decrypt_ok"] and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 128 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]      # whole-job: both ranks' batches
    legs = d["sharded"]["legs"]
    assert {"gate", "sample"} <= set(legs)
    for name, leg in legs.items():
        assert leg["rccl_ranks"] == 2 and leg["decrypt_ok"] and leg["value"] > 0 and leg["scaling"] == "strong", name
        assert leg["sample_groups"] * leg["gate_groups"] == 2
        assert any("blind_rotate" in k for k in leg["kernel_instantiations_rank0"])
    assert legs["gate"]["gate_groups"] == 2 and legs["gate"]["collectives_per_step"] > 0 and legs["gate"]["allgather_ms_rank0"] >= 0
    assert legs["sample"]["collectives_per_step"] == 0 and legs["sample"]["allgather_ms_rank0"] == 0
    assert d["sharded"]["choose_sharding"]["sample_groups"] * d["sharded"]["choose_sharding"]["gate_groups"] == 2
    assert d["sharded"]["params"]["security_bits_estimate"] >= 127.9           # the legs run the 128-bit set chosen for the circuit
